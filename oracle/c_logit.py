"""ctypes wrapper of the C restatement oracle/csrc/oracle_logit.c (oracle; test + cpu_baseline only).

Built by `__graft_entry__.build()` into oracle/_build/liboracle.so (git-ignored,
travels to the GPU box).  Validated against the NumPy restatement in
tests/test_oracle_models.py.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "liboracle.so")


def build_native() -> str:
    """Compile the restatement for THIS host (`gcc -O3 -march=native`, what PyTensor's C linker would do where the
    sampler runs) -- used by `bench.py`'s cpu_baseline leg on the GPU box.  The portable build
    (`__graft_entry__.build_oracle`, x86-64-v3) stays what tests and smoke() load.  Falls back to it if gcc fails."""
    import subprocess

    out = os.path.join(_DIR, "_build", "liboracle_native.so")
    src = os.path.join(_DIR, "csrc", "oracle_logit.c")
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", src, src.replace("oracle_logit.c", "oracle_logit_stat.c"), "-o", out, "-lmvec", "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out
    except Exception:
        return _SO


class CHierLogit:
    """``q -> (logp, grad)`` for a ModelSpec built by `pymc_amd.models.hier_logit`."""

    def __init__(self, spec, so_path=None, fn="oracle_hier_logit"):
        # fn="oracle_hier_logit_stat": the libmvec arrangement of the same formulas (oracle/csrc/oracle_logit_stat.c), for the
        # long statistical fixtures only
        lib = C.CDLL(so_path or _SO)
        self._fn = getattr(lib, fn)
        self._fn.restype = C.c_double
        self._fn.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        r = spec.logit_rows
        self.X = np.ascontiguousarray(r.X, dtype="float64")
        self.y = np.ascontiguousarray(r.y, dtype="int8")
        self.g = np.ascontiguousarray(r.group_idx, dtype="int32")
        self.N, self.D = self.X.shape
        self.n = spec.n
        self.G = (self.n - 2 * self.D) // self.D
        names = [v.value_name for v in spec.vars]
        assert names == ["mu", "sigma_log__", "z"], names

    def __call__(self, q):
        q = np.ascontiguousarray(q, dtype="float64")
        grad = np.empty(self.n)
        lp = self._fn(self.N, self.D, self.G, self.X.ctypes.data, self.y.ctypes.data, self.g.ctypes.data, q.ctypes.data, grad.ctypes.data)
        return lp, grad


class CRowsSpecLogpGrad:
    """``q -> (logp, grad)`` for ANY ModelSpec with a logit node: everything but the rows through the NumPy restatement
    (`oracle/ref_models.py`: priors of any family, further variables, Jacobians), the rows through the gcc loop
    `oracle_logit_rows` (the formulas of `ref_models._logit_rows`; pinned to it in tests/test_oracle_models.py).  What the
    generalised one-launch row passes are compared with at the benchmark's size, where the NumPy rows take seconds per call."""

    def __init__(self, spec, so_path=None):
        from oracle import ref_models

        self._ref = ref_models
        lib = C.CDLL(so_path or _SO)
        self._fn = lib.oracle_logit_rows
        self._fn.restype = C.c_double
        self._fn.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.spec = spec
        self.n = spec.n
        r = spec.logit_rows
        self.X = np.ascontiguousarray(r.X, dtype="float64")
        self.y = np.ascontiguousarray(r.y, dtype="int8")
        self.g = np.ascontiguousarray(r.group_idx, dtype="int32")
        self.calls = 0

    def _rows(self, spec, node, x):
        vm, vs, vz = (spec.vars[i] for i in (node.mu, node.sigma, node.z))
        D = vm.size
        mu, sg = x[vm.offset : vm.offset + D], x[vs.offset : vs.offset + D]
        z = x[vz.offset : vz.offset + vz.size].reshape(-1, D)
        G = z.shape[0]
        beta = np.ascontiguousarray(mu + sg * z)
        dbeta = np.empty((G, D))
        lp = self._fn(self.X.shape[0], D, G, self.X.ctypes.data, self.y.ctypes.data, self.g.ctypes.data, beta.ctypes.data, dbeta.ctypes.data)
        g = np.zeros(spec.n)
        g[vm.offset : vm.offset + D] = dbeta.sum(0)
        g[vs.offset : vs.offset + D] = (dbeta * z).sum(0)
        g[vz.offset : vz.offset + vz.size] = (dbeta * sg).ravel()
        return float(lp), g

    def set_extra_values(self, extra_vars):
        self._ref.SpecLogpGrad.set_extra_values(self, extra_vars)

    def __call__(self, q):
        self.calls += 1
        return self._ref.evaluate(self.spec, q, rows_fn=self._rows)
