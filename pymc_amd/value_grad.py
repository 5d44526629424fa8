"""`ValueGradFunction` stand-in backed by the HIP model kernels.

Replaces the compiled artefact of pymc/model/core.py:142-305: a callable
``f(q: float64[n]) -> (logp: float64 scalar, dlogp: float64[n])`` over the
unconstrained, raveled, concatenated value variables.  Exposes the attributes
the integrator reads (`_raveled_inputs`, `dtype`, `_extra_vars_shared`,
`_pytensor_function`; pymc/step_methods/hmc/integration.py:46-61,
pymc/step_methods/arraystep.py:205).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.linalg

from pymc_amd import _lib
from pymc_amd.blocking import DictToArrayBijection, RaveledVars
from pymc_amd.model_spec import ModelSpec


def _pack(spec: ModelSpec, rows_group_aligned: bool = True):
    """ModelSpec -> nuts_model_spec (+ the numpy buffers that must outlive the call)."""
    keep = []
    data = spec.data if getattr(spec, "n_device_data", None) is None else spec.data[: spec.n_device_data]   # (the rest: host-only constants of Deterministics)
    nv, nf, nd = len(spec.vars), len(spec.factors), len(data)
    vars_c = (_lib.Var * max(nv, 1))()
    for i, v in enumerate(spec.vars):
        vars_c[i] = _lib.Var(v.offset, v.size, v.transform, 0, v.lower, v.upper)
    fac_c = (_lib.Factor * max(nf, 1))()
    n_instr = sum(len(getattr(f, "prog", ())) for f in spec.factors)
    ins_c = (_lib.Instr * max(n_instr, 1))()
    ioff = 0
    for i, f in enumerate(spec.factors):
        fc = _lib.Factor()
        fc.dist, fc.size, fc.nargs, fc.konst = f.dist, f.size, len(f.args), f.konst
        for k, t in enumerate(f.args):
            for nm in ("a", "b", "c"):
                o = getattr(t, nm)
                setattr(fc.arg[k], nm, _lib.Operand(o.kind, max(o.ref, 0), o.c))
        prog = getattr(f, "prog", ())
        fc.n_instr, fc.instr_off = len(prog), ioff
        for ins in prog:
            z = getattr(ins, "z", None) or ins.y
            ins_c[ioff] = _lib.Instr(ins.op, 0, ins.k, _lib.Operand(ins.x.kind, max(ins.x.ref, 0), ins.x.c), _lib.Operand(ins.y.kind, max(ins.y.ref, 0), ins.y.c),
                                     _lib.Operand(z.kind, max(z.ref, 0), z.c))
            ioff += 1
        fac_c[i] = fc
    refs = (_lib.DataRef * max(nd, 1))()
    off = 0
    for i, d in enumerate(data):
        refs[i] = _lib.DataRef(off, d.size)
        off += d.size
    pool = np.concatenate(data).astype("float64") if nd else np.zeros(1)
    keep += [vars_c, fac_c, refs, pool, ins_c]
    s = _lib.ModelSpecC()
    s.instrs, s.n_instrs = (ins_c if n_instr else None), n_instr
    s.n_vars, s.n_factors, s.n_data = nv, nf, nd
    s.vars, s.factors, s.data = vars_c, fac_c, refs
    s.data_pool, s.data_pool_len = _lib.dptr(pool), off
    if spec.logit_rows is not None:
        r = spec.logit_rows
        X = np.ascontiguousarray(r.X, dtype="float64")
        y = np.ascontiguousarray(r.y, dtype="int8")
        g = np.ascontiguousarray(r.group_idx, dtype="int32")
        keep += [X, y, g]
        s.rows_N, s.rows_D = X.shape
        s.rows_G = spec.vars[r.z].size // X.shape[1]
        s.rows_X = _lib.dptr(X)
        s.rows_y = y.ctypes.data_as(C.POINTER(C.c_int8))
        s.rows_gid = g.ctypes.data_as(C.POINTER(C.c_int32))
        s.rows_mu, s.rows_sigma, s.rows_z = r.mu, r.sigma, r.z
        s.rows_opts = 0 if rows_group_aligned else 1   # NUTS_ROWS_NO_GROUP_ALIGNED
    if spec.mvnormal is not None:
        mv = spec.mvnormal
        # Cholesky + triangular solves, as quaddist_chol does (pymc/distributions/multivariate.py:165-185);
        # the device consumes the precision matrix so one leapfrog is a single symmetric mat-vec.
        L = scipy.linalg.cholesky(mv.cov, lower=True)
        prec = np.ascontiguousarray(scipy.linalg.cho_solve((L, True), np.eye(len(mv.mu))))
        prec = 0.5 * (prec + prec.T)
        mu = np.ascontiguousarray(mv.mu, dtype="float64")
        keep += [prec, mu]
        s.mvn_var, s.mvn_k = mv.var, len(mu)
        s.mvn_mu, s.mvn_prec = _lib.dptr(mu), _lib.dptr(prec)
        s.mvn_logdet = float(np.log(np.diag(L)).sum())
        if getattr(mv, "solver", "precision") == "cholesky":
            winv = np.ascontiguousarray(np.tril(scipy.linalg.solve_triangular(L, np.eye(len(mu)), lower=True)))
            keep.append(winv)
            s.mvn_winv = _lib.dptr(winv)
    mx = getattr(spec, "mixture_rows", None)
    if mx is not None:
        y = np.ascontiguousarray(mx.y, dtype="float64")
        keep.append(y)
        s.mix_N, s.mix_K, s.mix_mu = y.size, mx.K, mx.mu
        s.mix_y = _lib.dptr(y)
        s.mix_sigma = -1 if mx.sigma is None else mx.sigma
        s.mix_w_logits = -1 if mx.w_logits is None else mx.w_logits
        s.mix_assign = -1 if mx.assign is None else mx.assign
        if mx.sigma is None:
            sc = np.ascontiguousarray(mx.sigma_const, dtype="float64")
            keep.append(sc)
            s.mix_sigma_const = _lib.dptr(sc)
        if mx.w_logits is None:
            wc = np.ascontiguousarray(mx.w_const, dtype="float64")
            keep.append(wc)
            s.mix_w_const = _lib.dptr(wc)
        elif getattr(mx, "w_alpha", None) is not None:
            wa = np.ascontiguousarray(mx.w_alpha, dtype="float64")
            keep.append(wa)
            s.mix_w_simplex = 1
            s.mix_w_alpha = _lib.dptr(wa)
    gl = getattr(spec, "glm_rows", None)
    if gl is not None:
        X = np.ascontiguousarray(gl.X, dtype="float64")
        y = np.ascontiguousarray(gl.y, dtype="float64")
        keep += [X, y]
        s.glm_N, s.glm_P = X.shape
        s.glm_family = gl.family
        s.glm_beta = -1 if gl.beta is None else gl.beta
        s.glm_beta_derived = -1 if getattr(gl, "beta_derived", None) is None else gl.beta_derived
        s.glm_intercept = -1 if gl.intercept is None else gl.intercept
        s.glm_sigma = -1 if gl.sigma is None else gl.sigma
        s.glm_sigma_const = float(gl.sigma_const)
        s.glm_X, s.glm_y = _lib.dptr(X), _lib.dptr(y)
    lins = getattr(spec, "lins", None) or []
    if lins:
        lins_c = (_lib.Lin * len(lins))()
        for i, L in enumerate(lins):
            X = np.ascontiguousarray(L.X, dtype="float64")
            keep.append(X)
            lins_c[i].N, lins_c[i].P = X.shape
            lins_c[i].K = len(L.cols)
            lins_c[i].X = _lib.dptr(X)
            for k, (var, off_, stride) in enumerate(L.cols[:16]):
                lins_c[i].var[k], lins_c[i].off[k], lins_c[i].stride[k] = var, off_, stride
        keep.append(lins_c)
        s.n_lins, s.lins = len(lins), lins_c
    return s, keep


class DeviceValueGradFunction:
    """Device-resident logp/dlogp function of a :class:`ModelSpec`."""

    def __init__(self, spec: ModelSpec, device: int | None = None, rows_group_aligned: bool = True):
        lib = _lib.load()
        if device is not None:
            _lib.check(lib.nuts_set_device(int(device)), "nuts_set_device")
        self.spec = spec
        self.device = device
        self.dtype = "float64"
        self._raveled_inputs = True
        self._extra_vars_shared = {}
        self._extra_are_set = False
        self.trust_input = True
        self.rows_group_aligned_allowed = rows_group_aligned
        cspec, keep = _pack(spec, rows_group_aligned)
        _lib.sync_options_from_env()     # (tests / tools only: see _lib.py)
        self._handle = lib.nuts_model_create(C.byref(cspec))
        del keep
        if not self._handle:
            raise _lib.EngineError(f"nuts_model_create failed: {_lib.last_error()}")
        self.n = lib.nuts_model_ndim(self._handle)
        self._grad = np.empty(self.n)
        self._lp = np.empty(1)

    def model_scalar(self, name: str) -> float:
        out = C.c_double()
        _lib.check(_lib.load().nuts_model_get_scalar(self._handle, name.encode(), C.byref(out)), name)
        return out.value

    def bind_thread(self):
        """HIP's current device is per host thread: a worker thread that drives this model selects its device first."""
        if self.device is not None:
            _lib.check(_lib.load().nuts_set_device(int(self.device)), "nuts_set_device")

    # the integrator calls `_pytensor_function(q)` directly (integration.py:46-52)
    def _pytensor_function(self, q):
        q = np.ascontiguousarray(q, dtype="float64")
        grad = np.empty(self.n)
        lp = np.empty(1)
        rc = _lib.load().nuts_model_logp_grad(self._handle, _lib.dptr(q), _lib.dptr(lp), _lib.dptr(grad))
        _lib.check(rc, "nuts_model_logp_grad")
        return lp[0], grad

    def __call__(self, grad_vars, *, extra_vars=None):
        """`ValueGradFunction.__call__` (pymc/model/core.py:286-300)."""
        if extra_vars is not None:
            self.set_extra_values(extra_vars)
        elif self.spec.extra and not self._extra_are_set:
            raise ValueError("Extra values are not set.")
        if isinstance(grad_vars, RaveledVars):
            q = grad_vars.data
        elif isinstance(grad_vars, dict):
            q = DictToArrayBijection.map({v.value_name: grad_vars[v.value_name] for v in self.spec.vars}).data
        else:
            q = np.asarray(grad_vars[0] if isinstance(grad_vars, (list, tuple)) else grad_vars)
        return self._pytensor_function(q)

    def set_extra_values(self, extra_vars):
        """core.py:275-278: the non-gradient inputs are data vectors of the spec, rewritten on the device
        (`nuts_model_set_data`; values that did not change are not sent again)."""
        lib = _lib.load()
        ids, vals = [], []
        for name, idx in self.spec.extra.items():
            v = np.ascontiguousarray(np.asarray(extra_vars[name], dtype="float64").ravel())
            old = self._extra_vars_shared.get(name)
            if old is not None and old.shape == v.shape and np.array_equal(old, v):
                continue
            ids.append(idx)
            vals.append(v)
            self._extra_vars_shared[name] = v.copy()
        if ids:     # ONE call, stream-ordered (no host synchronisation per vector)
            ids_a = np.asarray(ids, dtype="int32")
            lens = np.asarray([v.size for v in vals], dtype="int64")
            flat = np.concatenate(vals) if len(vals) > 1 else vals[0]
            _lib.check(lib.nuts_model_set_data_many(self._handle, len(ids), ids_a.ctypes.data, _lib.dptr(flat), lens.ctypes.data), "nuts_model_set_data_many")
        self._extra_are_set = True

    def get_extra_values(self):  # core.py:280-284
        if self.spec.extra and not self._extra_are_set:
            raise ValueError("Extra values are not set.")
        return {name: self._extra_vars_shared[name].copy() for name in self.spec.extra}

    def time_kernels(self, q, reps=20):
        ms_tot, ms_dom = C.c_double(), C.c_double()
        q = np.ascontiguousarray(q, dtype="float64")
        rc = _lib.load().nuts_model_time_logp_grad(self._handle, _lib.dptr(q), reps, C.byref(ms_tot), C.byref(ms_dom))
        _lib.check(rc, "nuts_model_time_logp_grad")
        return ms_tot.value, ms_dom.value

    @property
    def algorithmic_bytes(self) -> int:
        return int(_lib.load().nuts_model_algorithmic_bytes(self._handle))

    def close(self):
        if getattr(self, "_handle", None):
            _lib.load().nuts_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
