"""Lockstep chains of one model on one GPU (BASELINE configs[2]: ``pm.sample(chains=4)`` on the MvNormal-2048 model).

The reference runs chains as independent workers (``pymc/sampling/mcmc.py:1385-1500``, ``sampling/parallel.py:477-589``); its
accelerator path advances them together (``jax.vmap`` over the chain axis, ``sampling/jax.py:341-348``).  `ChainGroup` is the
device-side form of the latter behind the unchanged step interface: every chain keeps its own step object, model handle, random
streams and host thread and is sampled by the ordinary calls; while the group exists the chains submit to one stream and the
leapfrog launches of chains that stand inside a tree at the same time become ONE launch that reads the precision matrix once
for all of them (``csrc/mvn_multi_kernel.h``, ``include/nuts_mi355.h`` "chain groups").  A chain in a group produces bitwise the
draws and statistics it produces alone (``tests/test_gpu_chain_group.py``).
"""

from __future__ import annotations

import ctypes as C
from typing import List, Optional

from pymc_amd import _lib


class ChainGroup:
    MAX_CHAINS = 4        # on the plain-fma kernels: a chain in the group is bitwise the chain alone
    MAX_CHAINS_ROWS = 8   # ... of the hierarchical-logit rows (csrc/rows_gal_kernel.h: one wave per chain, tiles shared through LDS), bitwise too
    MAX_CHAINS_WIDE = 16  # chains created under the options NUTS_MVN_ALIGNED = 8, NUTS_GROUP_WIDE = 1: the merged launch runs on the matrix
                          # cores (csrc/mvn_mfma_kernel.h); chains are then held to the oracle, not to bitwise equality with themselves alone

    def __init__(self, steps):
        lib = _lib.load()
        self._handle = lib.nuts_group_create()
        if not self._handle:
            raise _lib.EngineError(f"nuts_group_create failed: {_lib.last_error()}")
        self._steps = []
        try:
            for st in steps:
                _lib.check(lib.nuts_group_add(self._handle, st._chain), "nuts_group_add")
                self._steps.append(st)
        except Exception:
            self.close()
            raise

    @classmethod
    def try_create(cls, steps) -> Optional["ChainGroup"]:
        """The group, or None when the engine declines (a model that is not one MvNormal node on the row-aligned pass, a dense or
        host mass matrix, more than four chains): the chains then run as independent engines, as before."""
        if not 2 <= len(steps) <= cls.MAX_CHAINS_WIDE:
            return None
        try:
            return cls(steps)
        except (_lib.EngineError, ValueError):   # (NUTS_E_ARG surfaces as ValueError: the engine declined, with its reason)
            return None

    def launches(self) -> List[int]:
        """``[_, n1, n2, n3, n4]``: leapfrog launches submitted so far that carried 1, 2, 3, 4 chains."""
        out, cap = (C.c_int64 * 17)(), C.c_int32(0)
        _lib.check(_lib.load().nuts_group_launches_wide(self._handle, out, C.byref(cap)), "nuts_group_launches_wide")
        return [int(v) for v in out[: int(cap.value) + 1]]      # ([_, n1..n4], or [_, n1..n16] for a wide group)

    def mean_chains_per_launch(self) -> float:
        n = self.launches()
        tot = sum(n[1:])
        return sum(c * n[c] for c in range(1, len(n))) / tot if tot else 0.0

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _lib.load().nuts_group_destroy(self._handle)
            self._handle = None
            self._steps = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
