"""ctypes binding of libnuts_mi355.so (the C ABI in include/nuts_mi355.h).

There is no CPU fallback: if the shared library is missing or no HIP device is
visible the product path raises.  The library is built in-tree by
``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnuts_mi355.so")

NUTS_OK, NUTS_E_BAD_ENERGY, NUTS_E_ARG, NUTS_E_HIP, NUTS_E_LINALG, NUTS_E_CALLBACK = 0, 1, 2, 3, 4, 5


class EngineError(RuntimeError):
    """A HIP/runtime failure inside libnuts_mi355 (maps to ParallelSamplingError-style reporting)."""


class Operand(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ref", C.c_int32), ("c", C.c_double)]


class Term(C.Structure):
    _fields_ = [("a", Operand), ("b", Operand), ("c", Operand)]


class Instr(C.Structure):
    _fields_ = [("op", C.c_int32), ("pad", C.c_int32), ("k", C.c_double), ("x", Operand), ("y", Operand), ("z", Operand)]


class Factor(C.Structure):
    _fields_ = [
        ("dist", C.c_int32),
        ("size", C.c_int32),
        ("nargs", C.c_int32),
        ("n_instr", C.c_int32),
        ("konst", C.c_double),
        ("arg", Term * 4),
        ("instr_off", C.c_int32),
        ("pad", C.c_int32),
    ]


class Var(C.Structure):
    _fields_ = [
        ("offset", C.c_int32),
        ("size", C.c_int32),
        ("transform", C.c_int32),
        ("pad", C.c_int32),
        ("lower", C.c_double),
        ("upper", C.c_double),
    ]


class DataRef(C.Structure):
    _fields_ = [("offset", C.c_int64), ("size", C.c_int64)]


class Lin(C.Structure):     # nuts_lin
    _fields_ = [("N", C.c_int64), ("P", C.c_int32), ("K", C.c_int32), ("X", C.POINTER(C.c_double)),
                ("var", C.c_int32 * 16), ("off", C.c_int32 * 16), ("stride", C.c_int32 * 16)]


class ModelSpecC(C.Structure):
    _fields_ = [
        ("n_vars", C.c_int32),
        ("n_factors", C.c_int32),
        ("n_data", C.c_int32),
        ("pad", C.c_int32),
        ("vars", C.POINTER(Var)),
        ("factors", C.POINTER(Factor)),
        ("data", C.POINTER(DataRef)),
        ("data_pool", C.POINTER(C.c_double)),
        ("data_pool_len", C.c_int64),
        ("rows_N", C.c_int64),
        ("rows_D", C.c_int32),
        ("rows_G", C.c_int32),
        ("rows_X", C.POINTER(C.c_double)),
        ("rows_y", C.POINTER(C.c_int8)),
        ("rows_gid", C.POINTER(C.c_int32)),
        ("rows_mu", C.c_int32),
        ("rows_sigma", C.c_int32),
        ("rows_z", C.c_int32),
        ("rows_opts", C.c_int32),
        ("mvn_var", C.c_int32),
        ("mvn_k", C.c_int32),
        ("mvn_mu", C.POINTER(C.c_double)),
        ("mvn_prec", C.POINTER(C.c_double)),
        ("mvn_logdet", C.c_double),
        ("mvn_winv", C.POINTER(C.c_double)),
        ("instrs", C.POINTER(Instr)),
        ("n_instrs", C.c_int32),
        ("pad2", C.c_int32),
        ("mix_N", C.c_int64),
        ("mix_K", C.c_int32),
        ("mix_mu", C.c_int32),
        ("mix_sigma", C.c_int32),
        ("mix_w_logits", C.c_int32),
        ("mix_assign", C.c_int32),
        ("mix_w_simplex", C.c_int32),
        ("mix_y", C.POINTER(C.c_double)),
        ("mix_sigma_const", C.POINTER(C.c_double)),
        ("mix_w_const", C.POINTER(C.c_double)),
        ("mix_w_alpha", C.POINTER(C.c_double)),
        ("glm_N", C.c_int64),
        ("glm_P", C.c_int32),
        ("glm_family", C.c_int32),
        ("glm_beta", C.c_int32),
        ("glm_intercept", C.c_int32),
        ("glm_sigma", C.c_int32),
        ("glm_beta_derived", C.c_int32),
        ("glm_sigma_const", C.c_double),
        ("glm_X", C.POINTER(C.c_double)),
        ("glm_y", C.POINTER(C.c_double)),
        ("n_lins", C.c_int32),
        ("pad3", C.c_int32),
        ("lins", C.POINTER(Lin)),
    ]


class ChainConfig(C.Structure):
    _fields_ = [
        ("step_scale", C.c_double),
        ("Emax", C.c_double),
        ("target_accept", C.c_double),
        ("gamma", C.c_double),
        ("k", C.c_double),
        ("t0", C.c_double),
        ("adapt_step_size", C.c_int32),
        ("max_treedepth", C.c_int32),
        ("early_max_treedepth", C.c_int32),
        ("potential", C.c_int32),
        ("initial_mean", C.POINTER(C.c_double)),
        ("initial_diag", C.POINTER(C.c_double)),
        ("initial_weight", C.c_double),
        ("adaptation_window", C.c_int32),
        ("discard_window", C.c_int32),
        ("adaptation_window_multiplier", C.c_double),
        ("early_update", C.c_int32),
        ("pad", C.c_int32),
        ("dense_cov", C.POINTER(C.c_double)),
        ("dense_rand", C.POINTER(C.c_double)),
        ("exp_alpha", C.c_double),
        ("exp_stop_adaptation", C.c_double),
        ("exp_use_grads", C.c_int32),
        ("fa_update_window", C.c_int32),
    ]


class DrawStats(C.Structure):
    _fields_ = [
        ("depth", C.c_int64),
        ("step_size", C.c_double),
        ("mean_tree_accept", C.c_double),
        ("step_size_bar", C.c_double),
        ("tree_size", C.c_double),
        ("diverging", C.c_int32),
        ("reached_max_treedepth", C.c_int32),
        ("divergences", C.c_int64),
        ("energy_error", C.c_double),
        ("energy", C.c_double),
        ("max_energy_error", C.c_double),
        ("model_logp", C.c_double),
        ("process_time_diff", C.c_double),
        ("perf_counter_diff", C.c_double),
        ("perf_counter_start", C.c_double),
        ("index_in_trajectory", C.c_int64),
        ("n_uniforms_consumed", C.c_int32),
        ("warning", C.c_int32),
        ("divergence_energy_change", C.c_double),
        ("n_model_evals", C.c_int64),
    ]


class HmcStats(C.Structure):
    _fields_ = [
        ("step_size", C.c_double),
        ("step_size_bar", C.c_double),
        ("accept", C.c_double),
        ("energy_error", C.c_double),
        ("energy", C.c_double),
        ("model_logp", C.c_double),
        ("path_length", C.c_double),
        ("n_steps", C.c_int64),
        ("divergences", C.c_int64),
        ("diverging", C.c_int32),
        ("accepted", C.c_int32),
        ("process_time_diff", C.c_double),
        ("perf_counter_diff", C.c_double),
        ("perf_counter_start", C.c_double),
    ]


class Pcg64(C.Structure):
    _fields_ = [("state_hi", C.c_uint64), ("state_lo", C.c_uint64), ("inc_hi", C.c_uint64), ("inc_lo", C.c_uint64),
                ("has_uint32", C.c_int32), ("uinteger", C.c_uint32)]


class AdviConfig(C.Structure):
    _fields_ = [("N", C.c_int64), ("P", C.c_int32), ("family", C.c_int32), ("batch", C.c_int32), ("n_win", C.c_int32),
                ("sigma", C.c_double), ("prior_sd", C.c_double), ("learning_rate", C.c_double), ("epsilon", C.c_double),
                ("X", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double)), ("start", C.POINTER(C.c_double)),
                ("scale_cost_to_minibatch", C.c_int32), ("reserved0", C.c_int32)]


_PD = C.POINTER(C.c_double)
_VP = C.c_void_p

# callbacks of a host-owned potential (NUTS_POT_HOST; include/nuts_mi355.h)
VelocityFn = C.CFUNCTYPE(C.c_int, _VP, C.c_int32, _PD, _PD)
EnergyFn = C.CFUNCTYPE(C.c_int, _VP, C.c_int32, _PD, _PD, _PD)
VelocityEnergyFn = C.CFUNCTYPE(C.c_int, _VP, C.c_int32, _PD, _PD, _PD)

# every symbol include/nuts_mi355.h declares: (restype, argtypes)
SYMBOLS = {
    "nuts_device_count": (C.c_int, []),
    "nuts_set_device": (C.c_int, [C.c_int]),
    "nuts_last_error": (C.c_char_p, []),
    "nuts_model_create": (_VP, [C.POINTER(ModelSpecC)]),
    "nuts_model_destroy": (None, [_VP]),
    "nuts_model_ndim": (C.c_int32, [_VP]),
    "nuts_model_logp_grad": (C.c_int, [_VP, _PD, _PD, _PD]),
    "nuts_model_time_logp_grad": (C.c_int, [_VP, _PD, C.c_int, _PD, _PD]),
    "nuts_model_algorithmic_bytes": (C.c_int64, [_VP]),
    "nuts_model_debug_ticks": (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    "nuts_model_debug_tree": (C.c_int, [_VP, C.POINTER(C.c_int64), C.c_int64]),
    "nuts_model_get_scalar": (C.c_int, [_VP, C.c_char_p, _PD]),
    "nuts_chain_config_default": (None, [C.POINTER(ChainConfig)]),
    "nuts_chain_create": (_VP, [_VP, C.POINTER(ChainConfig)]),
    "nuts_chain_destroy": (None, [_VP]),
    "nuts_chain_reset_tuning": (C.c_int, [_VP]),
    "nuts_chain_set_tune": (C.c_int, [_VP, C.c_int]),
    "nuts_chain_set_iter_count": (C.c_int, [_VP, C.c_int64]),
    "nuts_chain_draw": (C.c_int, [_VP, _PD, _PD, _PD, C.c_int32, _PD, _PD, C.POINTER(DrawStats)]),
    "nuts_model_set_data": (C.c_int, [_VP, C.c_int32, _PD, C.c_int64]),
    "nuts_model_set_data_many": (C.c_int, [_VP, C.c_int32, _VP, _PD, _VP]),
    "nuts_chain_draw_many": (C.c_int, [_VP, _PD, _PD, _PD, C.c_int32, C.c_int32, _PD, C.POINTER(DrawStats), C.POINTER(C.c_int32)]),
    "nuts_chain_draw_hmc": (C.c_int, [_VP, _PD, _PD, _PD, C.c_double, C.c_int32, _PD, _PD, C.POINTER(HmcStats)]),
    "nuts_chain_leapfrog_test": (C.c_int, [_VP, _PD, _PD, C.c_double, C.c_int32, _PD, _PD, _PD]),
    "nuts_chain_state_size": (C.c_int64, [_VP]),
    "nuts_chain_get_state": (C.c_int, [_VP, _VP]),
    "nuts_chain_set_state": (C.c_int, [_VP, _VP]),
    "nuts_chain_get_scalar": (C.c_int, [_VP, C.c_char_p, _PD]),
    "nuts_chain_get_vector": (C.c_int, [_VP, C.c_char_p, _PD]),
    "nuts_chain_set_dense": (C.c_int, [_VP, _PD, _PD]),
    "nuts_chain_set_diag": (C.c_int, [_VP, _PD, _PD, _PD]),
    "nuts_chain_set_host_potential": (C.c_int, [_VP, VelocityFn, EnergyFn, VelocityEnergyFn, _VP]),
    "nuts_chain_welford_export": (C.c_int, [_VP, _VP]),
    "nuts_chain_welford_import": (C.c_int, [_VP, _VP]),
    "nuts_chain_set_log_step_bar": (C.c_int, [_VP, C.c_double, C.c_double]),
    "nuts_gibbs_plan": (C.c_int, [C.POINTER(Pcg64), C.c_int64, C.c_int32, _VP, _VP, _VP, _PD]),
    "nuts_gibbs_plan_shuffle": (C.c_int, [C.POINTER(Pcg64), C.c_int64, _VP]),
    "nuts_gibbs_plan_draws": (C.c_int, [C.POINTER(Pcg64), C.c_int64, _VP, _VP, _VP, _PD, C.POINTER(C.c_int32)]),
    "nuts_gibbs_plan_skip": (C.c_int, [C.POINTER(Pcg64), C.c_int64, C.c_int32]),
    "nuts_gibbs_create": (_VP, [C.c_int64, C.c_int32, _PD]),
    "nuts_gibbs_destroy": (None, [_VP]),
    "nuts_gibbs_stage_slots": (C.c_int, []),
    "nuts_gibbs_stage": (C.c_int, [_VP, C.c_int32, _VP, _VP, _PD]),
    "nuts_gibbs_sweep_staged": (C.c_int, [_VP, C.c_int32, _VP, _VP, C.c_int32, _PD, _PD, _PD, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _PD, _PD, _PD]),
    "nuts_gibbs_sweep": (C.c_int, [_VP, _VP, _PD, _PD, _PD, _VP, _VP, _PD, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _PD, _PD, _PD]),
    "nuts_gibbs_plan_doubles": (C.c_int, [C.POINTER(Pcg64), C.c_int64, C.c_int32, _VP, C.c_int64, _PD]),
    "nuts_gibbs_sweep_prop": (C.c_int, [_VP, _VP, _VP, _PD, _PD, _PD, _VP, _PD, _PD, _VP, C.POINTER(C.c_int64), _PD, _PD, _PD]),
    "nuts_set_option": (C.c_int, [C.c_char_p, C.c_int32]),
    "nuts_clear_options": (None, []),
    "nuts_advi_create": (_VP, [C.POINTER(AdviConfig)]),
    "nuts_advi_destroy": (None, [_VP]),
    "nuts_advi_steps": (C.c_int, [_VP, C.c_int32, _VP, _PD, _PD]),
    "nuts_advi_get_params": (C.c_int, [_VP, _PD, _PD]),
    "nuts_advi_set_params": (C.c_int, [_VP, _PD, _PD]),
    "nuts_chain_profile": (C.c_int, [_VP, C.c_int]),
    "nuts_chain_profile_read": (C.c_int, [_VP, _PD, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "nuts_group_create": (_VP, []),
    "nuts_group_add": (C.c_int, [_VP, _VP]),
    "nuts_group_remove": (C.c_int, [_VP, _VP]),
    "nuts_group_destroy": (None, [_VP]),
    "nuts_group_launches": (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    "nuts_group_launches_wide": (C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "nuts_unset_option": (C.c_int, [C.c_char_p]),
}

_lib = None


def _init_torch_runtime_first():
    """PyTorch wheels bundle their own copy of the HIP runtime (torch/lib/libamdhip64.so) while this library links
    the system one (/opt/rocm).  Both can live in one process, but only if torch's copy initialises FIRST; if the
    engine touched the GPU before `torch.cuda` did, torch later reports "no GPUs found" and the RCCL plumbing
    (trace gather, pooled adaptation) cannot start.  torch is plumbing for the collectives only, so it is brought up first only
    where it can matter: when the process has already imported it, when it runs under a distributed launcher (WORLD_SIZE > 1),
    or when asked to (PYMC_AMD_TORCH_FIRST=1: a process that will import torch LATER, e.g. the test suite).  A single-GPU
    user of the sampler never imports torch."""
    if os.environ.get("PYMC_AMD_SKIP_TORCH_INIT"):
        return
    import sys

    wanted = "torch" in sys.modules or os.environ.get("PYMC_AMD_TORCH_FIRST") == "1"
    try:
        wanted = wanted or int(os.environ.get("WORLD_SIZE", "1")) > 1
    except ValueError:
        pass
    if not wanted:
        return
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # torch absent or CPU-only box: the engine itself decides whether a GPU is usable
        pass


def load():
    """Load libnuts_mi355.so and bind every declared symbol; raise if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PYMC_AMD_LIB", LIB_PATH)   # (another build of the same library: A/B measurements)
    if not os.path.exists(path):
        raise EngineError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). pymc_amd has no CPU fallback."
        )
    _init_torch_runtime_first()
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def set_engine_option(name: str, value: int) -> None:
    """Select a non-default launch schedule for the models / chains created from now on (include/nuts_mi355.h, nuts_set_option)."""
    with _sync_lock:
        _synced.pop(name, None)      # (sync_options_from_env sets it again from the environment if it is there)
        check(load().nuts_set_option(name.encode(), int(value)), "nuts_set_option")


def unset_engine_option(name: str) -> None:
    with _sync_lock:
        _synced.pop(name, None)
        check(load().nuts_unset_option(name.encode()), "nuts_unset_option")


def sync_options_from_env() -> None:
    """TESTS AND tools/ ONLY.  With PYMC_AMD_HONOUR_NUTS_ENV=1 in the environment, the NUTS_* variables of the moment become the
    engine's schedule options (the engine itself never reads the environment).  Called when a model or chain is about to be
    created, so `monkeypatch.setenv("NUTS_ROWS_GA", "0")` in a test acts on the model the test creates next."""
    if os.environ.get("PYMC_AMD_HONOUR_NUTS_ENV") != "1":
        return
    lib = load()
    wanted = {}
    for k, v in os.environ.items():
        if k.startswith("NUTS_"):
            try:
                wanted[k] = int(v)
            except ValueError:
                pass
    # Chains are materialised from worker threads too: the table is never cleared (another thread creating a chain at that moment
    # would read defaults for options the environment sets) -- under one lock, the options that left the environment are unset, the
    # others (re)set; an option that did not change keeps its value throughout.
    with _sync_lock:
        for k in [k for k in _synced if k not in wanted]:
            lib.nuts_unset_option(k.encode())
            del _synced[k]
        for k, v in wanted.items():
            if _synced.get(k) != v:
                lib.nuts_set_option(k.encode(), v)
                _synced[k] = v


_sync_lock = threading.Lock()
_synced: dict = {}


def last_error() -> str:
    return load().nuts_last_error().decode()


def dptr(a: np.ndarray):
    return a.ctypes.data_as(_PD)


def check(rc: int, what: str = ""):
    if rc == NUTS_OK:
        return
    msg = last_error()
    if rc == NUTS_E_BAD_ENERGY:
        from pymc_amd.exceptions import SamplingError

        raise SamplingError(f"Bad initial energy: {msg}")
    if rc == NUTS_E_ARG:
        raise ValueError(f"{what}: {msg}")
    raise EngineError(f"{what}: {msg} (code {rc})")
