"""Random element-wise graphs through the op-by-op lowering (pymc_amd/lowering.py `_general`): expression DAGs over scalar and vector
variables, data vectors, broadcasts between them, gathers, short reductions, comparisons and switches -- built with the graph
stand-in of tests/stubgraph.py as `pm.Potential` terms (model/core.py:2007-2124) -- lowered to expression programs and evaluated by
the oracle's interpreter (value + reverse sweep), against torch autograd of the SAME graph (tests/graph_torch.py).  What the
lowering does to a graph on the way -- constant folding, common sub-expressions, index push-down, unrolled reductions, operands
turned into gathers -- must not change its value or its gradient.  Host only: the device runs the same programs through the same
opcodes (tests/test_general_lowering.py holds it to the same autograd values)."""
import numpy as np
import pytest

import graph_torch as gt
import stubgraph as sg
from oracle import ref_models
from pymc_amd.lowering import NotLowerable, lower_to_spec

pt = sg.pt
N = 5
IDX = np.array([3, 0, 0, 4, 1])
DATA = np.array([0.3, -1.2, 0.8, 2.0, -0.4])
POS = np.array([0.5, 1.7, 0.9, 2.4, 1.1])


def _leaf(rng, env, want_vec):
    pool = env["vec"] if want_vec else env["sca"]
    return pool[rng.integers(len(pool))]


def _expr(rng, env, depth, want_vec):
    """A random expression of bounded magnitude: (graph, is_vector)."""
    if depth == 0 or rng.uniform() < 0.15:
        return _leaf(rng, env, want_vec)
    kind = rng.integers(0, 20)
    sub = lambda v=want_vec: _expr(rng, env, depth - 1, v)                       # noqa: E731
    mixed = lambda: _expr(rng, env, depth - 1, want_vec and rng.uniform() < 0.6)  # noqa: E731  (a scalar operand broadcasts)
    if kind == 0:
        return sub() + mixed()
    if kind == 1:
        return sub() - mixed()
    if kind == 2:
        return sub() * pt.tanh(mixed())
    if kind == 3:
        return sub() / (1.0 + pt.sqr(mixed()))
    if kind == 4:
        return pt.exp(0.3 * pt.tanh(sub()))
    if kind == 5:
        return pt.log1p(pt.sqr(sub()))
    if kind == 6:
        return pt.sqrt(1.0 + pt.sqr(sub()))
    if kind == 7:
        return pt.sigmoid(sub())
    if kind == 8:
        return pt.softplus(sub())
    if kind == 9:
        return pt.abs(sub() - 0.123)
    if kind == 10:
        return pt.maximum(sub(), mixed())
    if kind == 11:
        return pt.minimum(sub(), mixed())
    if kind == 12:
        return pt.switch(pt.gt(sub(), mixed()), sub(), mixed() * 0.5)
    if kind == 13:
        return pt.pow(1.0 + pt.sqr(sub()), 0.3 * pt.tanh(mixed()))
    if kind == 14:
        return pt.gammaln(1.5 + pt.sqr(sub()))
    if kind == 15:
        return pt.erf(sub()) + pt.log(1.0 + pt.sqr(mixed()))
    if kind == 16 and want_vec:
        return _expr(rng, env, depth - 1, True)[IDX]                                 # a gather of an expression
    if kind == 17 and not want_vec:
        return pt.sum(_expr(rng, env, depth - 1, True), axis=0)                     # a short reduction inside an expression
    if kind == 18:
        return pt.clip(sub(), -0.7, 1.3)
    if kind == 19:
        return pt.switch(pt.and_(pt.ge(sub(), -0.2), pt.le(mixed(), 0.9)), sub(), -sub())
    return sub() * 0.7 + 0.1


def _model(seed):
    rng = np.random.default_rng(seed)
    m = sg.StubModel()
    a = m.Normal("a", 0.0, 1.0)
    b = m.HalfNormal("b", 1.0)
    v = m.Normal("v", 0.0, 1.0, shape=(N,))
    u = m.Beta("u", 2.0, 2.0, shape=(N,))
    env = {"sca": [a, b, sg.as_tensor(0.37)], "vec": [v, u, sg.as_tensor(DATA), sg.as_tensor(POS), v[IDX]]}
    for k in range(3):
        m.Potential(f"pot{k}", _expr(rng, env, 4, bool(k % 2 == 0)))
    return m


@pytest.mark.parametrize("seed", range(150))
def test_random_graphs_lower_to_programs_that_keep_value_and_gradient(seed):
    m = _model(seed)
    try:
        spec = lower_to_spec(m)
    except NotLowerable as e:      # (a program longer than the IR's 128 instructions is a refusal, not an error)
        assert "instruction" in str(e) or "MAX_FACTOR_INSTR" in str(e) or "longer" in str(e), str(e)
        pytest.skip(f"refused: {e}")
    n = spec.n
    rng = np.random.default_rng(1000 + seed)
    # (not at q = 0: every variable equal means exact ties of DIFFERENT variables in maximum / minimum, where conventions differ --
    # torch splits the gradient, PyTensor's `ScalarMaximum.L_op` credits the first operand, which is what the lowering emits.  Ties
    # of the SAME quantity reached along two paths -- `minimum(switch(c, a, b), a)` -- are not exotic and are covered: this test
    # found the opcode's own reverse rule crediting both operands there, twice the gradient)
    for scale in (0.3, 0.6, 1.0):
        q = rng.normal(size=n) * scale
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert np.isfinite(lp0)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (seed, scale, lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (seed, scale, float(np.max(np.abs(g - g0))))


# ---- random MODELS: priors and likelihoods of the template families with parameters that are numbers, earlier variables or
# expressions of them -- the template path (`_Lowering._factor`: unification against the reference's own `logp` forms, the Jacobian
# terms of the value transforms, fast flags, the fall-back to the op-by-op path when a template's arguments are outside the affine
# terms) against torch autograd of the graphs the reference's code built ----------------------------------------------------------
_YR = np.array([0.3, -0.7, 1.9, 0.4, -1.1, 0.8])
_YP = np.array([0.6, 1.4, 0.2, 2.7, 0.9, 1.1])
_YU = np.array([0.12, 0.55, 0.81, 0.33, 0.67, 0.45])
_YC = np.array([0.0, 3.0, 1.0, 2.0, 5.0, 1.0])
_YB = np.array([1.0, 0.0, 1.0, 1.0, 0.0, 0.0])
_G6 = np.array([2, 0, 1, 1, 2, 0])


def _dist_model(seed):
    rng = np.random.default_rng(7000 + seed)
    m = sg.StubModel()
    real, pos, unit = [], [], []            # variables (and expressions) by support

    vec_ok = [False]        # (a scalar variable takes scalar parameters: with a vector parameter PyMC would make the variable a vector)

    def pick(pool, const):
        pool = [v for v in pool if vec_ok[0] or not v.type.shape]
        if pool and rng.uniform() < 0.7:
            return pool[rng.integers(len(pool))]
        return const

    def real_param():
        r = pick(real, float(np.round(rng.normal(), 2)))
        if isinstance(r, sg.Variable) and pos and rng.uniform() < 0.4:
            return r + 0.5 * pick(pos, 1.0) * float(np.round(rng.normal(), 2))      # an affine / product term
        return r

    def pos_param():
        p = pick(pos, float(np.round(rng.uniform(0.4, 2.5), 2)))
        if isinstance(p, sg.Variable) and rng.uniform() < 0.3:
            return m.math.exp(0.3 * pick(real, 0.2)) * p if real else 1.5 * p
        return p

    n_free = int(rng.integers(3, 7))
    for i in range(n_free):
        kind = rng.integers(0, 14)
        shape = (3,) if rng.uniform() < 0.3 else ()
        vec_ok[0] = bool(shape)
        nm = f"x{i}"
        if kind == 11:        # interval transform with constant bounds
            lo = float(np.round(rng.uniform(-2.0, 0.0), 2))
            (pos if lo >= 0 else real).append(m.Uniform(nm, lo, lo + float(np.round(rng.uniform(1.0, 3.0), 2)), shape=shape))
            continue
        if kind == 12:        # TruncatedNormal between constant bounds: interval transform over the bound arguments
            lo = float(np.round(rng.uniform(-1.5, 0.0), 2))
            up = lo + float(np.round(rng.uniform(1.5, 3.0), 2))
            real.append(m.TruncatedNormal(nm, real_param(), pos_param(), lower=lo, upper=up, shape=shape))
            continue
        if kind == 13:
            unit.append(m.Uniform(nm, 0.0, 1.0, shape=shape))
            continue
        if kind == 0:
            real.append(m.Normal(nm, real_param(), pos_param(), shape=shape))
        elif kind == 1:
            pos.append(m.HalfNormal(nm, pos_param(), shape=shape))
        elif kind == 2:
            pos.append(m.HalfCauchy(nm, pos_param(), shape=shape))
        elif kind == 3:
            pos.append(m.Exponential(nm, pos_param(), shape=shape))
        elif kind == 4:
            pos.append(m.Gamma(nm, float(np.round(rng.uniform(1.2, 3.0), 2)), pos_param(), shape=shape))
        elif kind == 5:
            unit.append(m.Beta(nm, float(np.round(rng.uniform(1.2, 3.0), 2)), float(np.round(rng.uniform(1.2, 3.0), 2)), shape=shape))
        elif kind == 6:
            real.append(m.StudentT(nm, float(np.round(rng.uniform(2.5, 8.0), 1)), real_param(), pos_param(), shape=shape))
        elif kind == 7:
            real.append(m.Laplace(nm, real_param(), pos_param(), shape=shape))
        elif kind == 8:
            real.append(m.Cauchy(nm, real_param(), pos_param(), shape=shape))
        elif kind == 9:
            pos.append(m.LogNormal(nm, real_param(), pos_param(), shape=shape))
        else:
            pos.append(m.InverseGamma(nm, float(np.round(rng.uniform(2.2, 4.0), 2)), pos_param(), shape=shape))
    vec_ok[0] = False
    scal = lambda pool: [v for v in pool if not v.type.shape]              # noqa: E731  (likelihood parameters: scalars broadcast)
    vecs = [v for v in real if v.type.shape]
    for j in range(int(rng.integers(1, 4))):
        kind = rng.integers(0, 10)
        nm = f"y{j}"
        if kind == 7 and vecs:      # varying intercepts: a vector variable gathered by a constant index, plus a scalar
            m.Normal(nm, vecs[rng.integers(len(vecs))][_G6] + pick(scal(real), 0.3), pick(scal(pos), 0.9), observed=_YR)
            continue
        if kind == 8:               # a logit link written out
            m.Bernoulli(nm, pick(scal(real), 0.2) + pick(scal(real), -0.4) * sg.as_tensor(_YR), observed=_YB)
            continue
        if kind == 9 and scal(unit):
            m.Binomial(nm, 7.0, scal(unit)[rng.integers(len(scal(unit)))], observed=_YC)
            continue
        kind = kind % 7
        def rp():      # a likelihood's location: a number, a scalar variable, or a vector variable gathered per observation
            vr = [v for v in real if v.type.shape]
            if vr and rng.uniform() < 0.3:
                return vr[rng.integers(len(vr))][_G6]
            return pick(scal(real), float(np.round(rng.normal(), 2)))

        def pp():      # ... its scale / rate likewise
            vp = [v for v in pos if v.type.shape]
            if vp and rng.uniform() < 0.3:
                return vp[rng.integers(len(vp))][_G6]
            return pick(scal(pos), float(np.round(rng.uniform(0.5, 2.0), 2)))
        if kind == 0:
            m.Normal(nm, rp(), pp(), observed=_YR)
        elif kind == 1:
            m.StudentT(nm, 4.0, rp(), pp(), observed=_YR)
        elif kind == 2:
            m.Gamma(nm, float(np.round(rng.uniform(1.5, 3.0), 2)), pp(), observed=_YP)
        elif kind == 3:
            m.Poisson(nm, pp(), observed=_YC)
        elif kind == 4:
            m.LogNormal(nm, rp(), pp(), observed=_YP)
        elif kind == 5:
            m.Laplace(nm, rp(), pp(), observed=_YR)
        else:
            m.Exponential(nm, pp(), observed=_YP)
    return m


@pytest.mark.parametrize("seed", range(120))
def test_random_models_of_the_template_families_keep_value_and_gradient(seed):
    m = _dist_model(seed)
    spec = lower_to_spec(m)
    rng = np.random.default_rng(2000 + seed)
    for scale in (0.0, 0.4, 0.8):
        q = rng.normal(size=spec.n) * scale
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert np.isfinite(lp0)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (seed, scale, lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (seed, scale, float(np.max(np.abs(g - g0))))


# ---- random WRITINGS of a regression: the linear predictor as `pm.math.dot(X, beta)`, `X @ beta`, written out column by column, with
# the intercept before or after, coefficients that are a variable or an expression of variables (non-centred), three families -- the
# GLM node's recognition (`_Lowering._glm`) and its derived-vector form against torch autograd of the graph ----------------------
def _glm_model(seed):
    rng = np.random.default_rng(9000 + seed)
    P, N = int(rng.integers(2, 7)), 14
    X = np.round(rng.normal(size=(N, P)), 3)
    m = sg.StubModel()
    how = rng.integers(0, 3)
    if how == 0:
        beta = m.Normal("beta", 0.0, 1.5, shape=(P,))
    elif how == 1:                               # non-centred with scalar hyper-parameters
        mu = m.Normal("mu", 0.0, 1.0)
        tau = m.HalfNormal("tau", 1.0)
        beta = mu + tau * m.Normal("z", 0.0, 1.0, shape=(P,))
    else:                                        # ... with a hyper-parameter per coefficient
        mu = m.Normal("mu", 0.0, 1.0, shape=(P,))
        tau = m.HalfNormal("tau", 1.0, shape=(P,))
        beta = mu + tau * m.Normal("z", 0.0, 1.0, shape=(P,))
    writing = rng.integers(0, 3)
    if writing == 0:
        eta = m.math.dot(sg.as_tensor(X), beta)
    elif writing == 1:
        eta = sg.as_tensor(X) @ beta
    elif how == 0:                               # written out (elements of a VARIABLE: `beta[0] * x0 + beta[1] * x1 + ...`)
        eta = beta[0] * sg.as_tensor(X[:, 0])
        for k in range(1, P):
            eta = eta + beta[k] * sg.as_tensor(X[:, k])
    else:
        eta = m.math.dot(sg.as_tensor(X), beta)
    icpt = rng.integers(0, 3)
    if icpt == 1:
        a = m.Normal("alpha", 0.0, 2.0)
        eta = a + eta if rng.uniform() < 0.5 else eta + a
    elif icpt == 2:
        eta = eta + 0.35
    fam = rng.integers(0, 3)
    if fam == 0:
        s_ = m.HalfNormal("s", 1.0) if rng.uniform() < 0.6 else 0.8
        m.Normal("y", eta, s_, observed=np.round(rng.normal(size=N), 3))
    elif fam == 1:
        m.Bernoulli("y", eta, observed=(rng.uniform(size=N) < 0.5).astype("float64"))
    else:
        m.Poisson("y", m.math.exp(0.3 * eta) if rng.uniform() < 0.5 else m.math.exp(eta), observed=rng.poisson(2.0, size=N).astype("float64"))
    return m


@pytest.mark.parametrize("seed", range(80))
def test_random_writings_of_a_regression_keep_value_and_gradient(seed):
    m = _glm_model(seed)
    spec = lower_to_spec(m)
    rng = np.random.default_rng(3000 + seed)
    for scale in (0.0, 0.3, 0.6):
        q = rng.normal(size=spec.n) * scale
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert np.isfinite(lp0)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (seed, scale, lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (seed, scale, float(np.max(np.abs(g - g0))))


# ---- shapes: matrices built by broadcasting a vector [5] against a vector [3], reduced along either axis (sum, max, logsumexp),
# softmax rows, columns picked by slices, stacks and concatenations, cumulative sums, piecewise assembly (`set_subtensor`) -- the
# shape-aware half of the op-by-op lowering (broadcasts -> gathers, index push-down, unrolled reductions) --------------------------
W3 = np.array([0.7, -0.4, 1.2])


def _shape_model(seed):
    rng = np.random.default_rng(11000 + seed)
    m = sg.StubModel()
    v = m.Normal("v", 0.0, 1.0, shape=(N,))
    w = m.Normal("w", 0.0, 1.0, shape=(3,))
    s = m.HalfNormal("s", 1.0)
    col = lambda x: x[:, None]          # noqa: E731
    row = lambda x: x[None, :]          # noqa: E731

    def mat():
        k = rng.integers(0, 6)
        if k == 0:
            return col(v) * row(w)
        if k == 1:
            return col(v) + row(sg.as_tensor(W3)) * s
        if k == 2:
            return pt.tanh(col(sg.as_tensor(DATA)) - row(w))
        if k == 3:
            return col(pt.sigmoid(v)) * row(pt.exp(0.3 * w))
        if k == 4:
            return pt.sqr(col(v) - row(w)) * 0.5
        return col(v) * row(sg.as_tensor(W3)) + s

    def vec5():
        k = rng.integers(0, 8)
        M = mat()
        if k == 0:
            return M.sum(axis=1)
        if k == 1:
            return pt.logsumexp(M, axis=1)
        if k == 2:
            return pt.max(M, axis=1) if hasattr(pt, "max") else M.sum(axis=1)
        if k == 3:
            return M[:, int(rng.integers(0, 3))]
        if k == 4:
            return pt.log(pt.softmax(M, axis=-1)[:, int(rng.integers(0, 3))])
        if k == 5:
            return (M[:, 1:] - M[:, :-1]).sum(axis=1)
        if k == 6:
            return pt.cumsum(pt.softplus(M), axis=1)[:, 2]
        return (M * row(sg.as_tensor(W3))).sum(axis=-1)

    def vec3():
        k = rng.integers(0, 5)
        if k == 0:
            return mat().sum(axis=0)
        if k == 1:
            return pt.cumsum(pt.exp(0.2 * w), axis=0)
        if k == 2:
            return pt.stack([w[0], w[1] * s, pt.tanh(w[2])])
        if k == 3:
            x = pt.set_subtensor(pt.empty((3,))[0:1], w[0:1])
            return pt.set_subtensor(x[1:], pt.exp(0.3 * w[1:]))
        return pt.concatenate([w[:2], sg.as_tensor(np.array([0.25]))])

    m.Potential("p5", pt.tanh(vec5()) + 0.1 * vec5())
    m.Potential("p3", pt.sqr(vec3()) * -0.5)
    m.Potential("p1", pt.sum(vec3() * sg.as_tensor(W3), axis=0) + pt.sum(pt.tanh(vec5()), axis=0))
    return m


@pytest.mark.parametrize("seed", range(100))
def test_random_shapes_keep_value_and_gradient(seed):
    m = _shape_model(seed)
    try:
        spec = lower_to_spec(m)
    except NotLowerable as e:
        assert "instruction" in str(e), str(e)
        pytest.skip(f"refused: {e}")
    rng = np.random.default_rng(4000 + seed)
    for scale in (0.3, 0.6, 1.0):
        q = rng.normal(size=spec.n) * scale
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert np.isfinite(lp0)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (seed, scale, lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (seed, scale, float(np.max(np.abs(g - g0))))


def test_censored_likelihoods_lower_op_by_op():
    """`pm.Censored(name, Normal.dist(mu, sigma), lower, upper, observed=y)` (distributions/censored.py; logprob/censoring.py:198-250
    `clip_logprob` over the base distribution's own `logp`, `logcdf`, `logccdf` -- executed from the reference by tests/stubgraph.py):
    two-sided, right-censored, and an Exponential whose logccdf is `log1mexp(logcdf)`.  Host only (see the module docstring)."""
    rng = np.random.default_rng(0)
    raw = rng.normal(0.5, 1.2, size=20)

    def both(m):
        m.Censored("y", ("Normal", dict(mu=m.Normal("mu", 0.0, 2.0), sigma=m.HalfNormal("s", 1.0))), -0.5, 1.5, observed=np.clip(raw, -0.5, 1.5))

    def right(m):
        m.Censored("y", ("Normal", dict(mu=m.Normal("mu", 0.0, 2.0), sigma=m.HalfNormal("s", 1.0))), None, 1.5, observed=np.minimum(raw, 1.5))

    def expo(m):
        m.Censored("y", ("Exponential", dict(lam=m.HalfNormal("l", 1.0))), None, 2.0, observed=np.minimum(rng.exponential(1.0, size=15), 2.0))

    for build in (both, right, expo):
        m = sg.StubModel()
        build(m)
        spec = lower_to_spec(m)
        assert spec.factors[-1].prog            # (no template: the graph itself is the program)
        for scale in (0.0, 0.2, 0.4):
            q = np.random.default_rng(3).normal(size=spec.n) * scale
            lp0, g0 = gt.joint_logp_grad(m, q)
            lp, g = ref_models.evaluate(spec, q)
            # (far in the tails autograd multiplies the zero adjoint of an unselected `switch` branch by an infinite derivative -- NaN, in
            # torch as in PyTensor; the interpreter does not propagate zero adjoints.  Compared where autograd is finite.)
            assert np.all(np.isfinite(g0))
            assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)) and np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0)))


# ---- the same random graphs as `pm.Deterministic`s (model/core.py:1940-2005): what the trace records (backends/base.py:183-191) is the
# graph's value at the draw -- the lowered program evaluated by the host at the constrained values (pymc_amd/model_spec.py
# `eval_program`), against torch evaluation of the graph at the raw point --------------------------------------------------------------
def _recorded(m, spec, qs):
    from pymc_amd.backends import NDArray

    tr = NDArray(model=spec)
    tr.setup(len(qs), 0)
    tr.record_batch(np.asarray(qs), None)
    torch = gt._torch()
    want = {}
    for name, var in m.deterministics.items():
        rows = []
        for q in qs:
            qt = torch.tensor(q, dtype=torch.float64)
            values, off = {}, 0
            for v in m.value_vars:
                shp = tuple(m.value_shapes[v.name])
                size = int(np.prod(shp)) if shp else 1
                values[id(v)] = qt[off:off + size].reshape(shp)
                off += size
            rows.append(np.asarray(gt.evaluate(var, values, {}).to(torch.float64).numpy(), dtype="float64").ravel())
        want[name] = np.stack(rows)
    return tr, want


@pytest.mark.parametrize("seed", range(60))
def test_random_graphs_as_deterministics_are_recorded_with_the_graphs_values(seed, caplog):
    rng = np.random.default_rng(21000 + seed)
    if seed % 2:
        m = _shape_model(seed)
        v, w, s = (r.expr for r in m.free)
        d = {"d_cum": pt.cumsum(pt.exp(0.2 * w), axis=0), "d_outer": (v[:, None] * w[None, :]).sum(axis=1) + s,
             "d_piece": pt.set_subtensor(pt.set_subtensor(pt.empty((3,))[0:1], w[0:1])[1:], pt.exp(0.3 * w[1:])),
             "d_soft": pt.softmax(v[:, None] * sg.as_tensor(W3)[None, :] + s, axis=-1)[:, int(rng.integers(0, 3))],
             "d_cat": pt.concatenate([w[:2] * s, sg.as_tensor(np.array([0.25]))])}
    else:
        m = _model(seed)
        a, b, v, u = (r.expr for r in m.free)
        env = {"sca": [a, b, sg.as_tensor(0.37)], "vec": [v, u, sg.as_tensor(DATA), sg.as_tensor(POS), v[IDX]]}
        d = {f"d{k}": _expr(rng, env, 4, bool(k % 2 == 0)) for k in range(3)}
    for name, g in d.items():
        m.Deterministic(name, g)
    try:
        with caplog.at_level("WARNING", logger="pymc_amd"):
            spec = lower_to_spec(m)
    except NotLowerable as e:
        assert "instruction" in str(e) or "longer" in str(e), str(e)
        pytest.skip(f"refused: {e}")
    left_out = [r.getMessage() for r in caplog.records if "not recorded" in r.getMessage()]
    assert all("instruction" in t or "longer" in t for t in left_out), left_out      # (only the program-length limit may leave one out)
    qs = np.random.default_rng(5000 + seed).normal(size=(3, spec.n)) * 0.7
    tr, want = _recorded(m, spec, qs)
    assert len(spec.deterministics) + len(left_out) == len(d)
    for name in spec.deterministics:
        got = tr.samples[name].reshape(len(qs), -1)
        np.testing.assert_allclose(got, np.broadcast_to(want[name], got.shape), rtol=1e-12, atol=1e-13, err_msg=f"{seed} {name}")


# ---- random models over the structured priors: zero-sum vectors (multivariate.py:2654-2807), Gaussian random walks and autoregressions
# (timeseries.py) of random lengths and orders, used through gathers, element-wise and under short reductions ------------------------------
def _structured_model(seed):
    rng = np.random.default_rng(31000 + seed)
    m = sg.StubModel()
    a = m.Normal("a", 0.0, 1.0)
    s = m.HalfNormal("s", 1.0)
    K = int(rng.integers(2, 13))
    z = m.ZeroSumNormal("z", sigma=s if rng.uniform() < 0.5 else float(rng.uniform(0.3, 2.0)), shape=(K,))
    n_obs = int(rng.integers(K, 3 * K + 2))
    g = rng.integers(0, K, size=n_obs)
    kind = rng.integers(0, 4)
    if kind == 0:
        m.Normal("y", mu=a + z[g], sigma=0.6, observed=rng.normal(size=n_obs))
    elif kind == 1:
        m.Poisson("y", pt.exp(0.3 * a + 0.5 * z), observed=rng.poisson(2.0, size=K).astype("float64"))
    elif kind == 2:
        m.Bernoulli("y", logit_p=a + z[g] * s, observed=(rng.uniform(size=n_obs) < 0.5).astype("float64"))
    else:
        m.Potential("pz", -0.5 * pt.sqr(pt.sum(pt.tanh(z) * sg.as_tensor(rng.normal(size=K)), axis=0) - a))
    T = int(rng.integers(4, 40))
    if rng.uniform() < 0.5:
        w = m.GaussianRandomWalk("w", mu=0.1 * a if rng.uniform() < 0.5 else 0.0, sigma=s if rng.uniform() < 0.6 else 0.5,
                                 init_dist=("Normal", dict(mu=0.0, sigma=float(rng.uniform(0.5, 3.0)))), shape=(T,))
    else:
        p = int(rng.integers(1, min(4, T - 1)))
        const = bool(rng.uniform() < 0.5)
        rho = m.Normal("rho", 0.0, 0.4, shape=(p + int(const),))
        w = m.AR("w", rho, sigma=s if rng.uniform() < 0.6 else 0.5, init_dist=("Normal", dict(mu=0.0, sigma=1.5)), constant=const, shape=(T,))
    if rng.uniform() < 0.5:
        m.Normal("yw", mu=w, sigma=0.4, observed=rng.normal(size=T))
    else:
        m.StudentT("yw", 4.0, mu=0.0, sigma=pt.exp(0.3 * w), observed=rng.normal(size=T))
    return m


@pytest.mark.parametrize("seed", range(60))
def test_random_models_over_zero_sum_vectors_and_time_series_keep_value_and_gradient(seed):
    m = _structured_model(seed)
    spec = lower_to_spec(m)
    rng = np.random.default_rng(6000 + seed)
    for scale in (0.3, 0.8):
        q = rng.normal(size=spec.n) * scale
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert np.isfinite(lp0)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (seed, scale, lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (seed, scale, float(np.max(np.abs(g - g0))))
    assert "z" in spec.deterministics


# ---- matrix products over a short inner dimension, outside the dense nodes: [N, P] @ [P], [N, P] @ [P, K], [P] @ [P, K]; either operand
# a constant, a variable or an expression; the result under an element-wise function, a softmax row, a reduction ---------------------------
def _dot_model(seed):
    rng = np.random.default_rng(41000 + seed)
    m = sg.StubModel()
    Nr, P, K = int(rng.integers(3, 9)), int(rng.integers(1, 5)), int(rng.integers(2, 4))      # (P K terms per row: the programs' 128 instructions)
    X = sg.as_tensor(rng.normal(size=(Nr, P)))
    B = m.Normal("B", 0.0, 1.0, shape=(P, K))
    b = m.Normal("b", 0.0, 1.0, shape=(P,))
    s = m.HalfNormal("s", 1.0)
    kind = rng.integers(0, 6)
    if kind == 0:
        eta = pt.dot(X, B)                                              # [N, K]
    elif kind == 1:
        eta = pt.dot(X * s, pt.tanh(B))                                 # expressions on both sides
    elif kind == 2:
        eta = pt.dot(X, B) + pt.dot(X, b)[:, None]                      # a mat-mat next to a mat-vec
    elif kind == 3:
        eta = pt.dot(pt.tanh(X + s), B * 0.5)
    elif kind == 4:
        eta = pt.dot(X, B) * pt.dot(b, B)[None, :]                      # [P] @ [P, K] -> [K]
    else:
        eta = pt.dot(X, pt.exp(0.2 * B)) - s
    use = rng.integers(0, 4)
    if use == 0:
        yc = rng.integers(0, K, size=Nr).astype("float64")
        m.Categorical("y", p=pt.softmax(eta, axis=-1), observed=yc)
    elif use == 1:
        m.Potential("p", -0.5 * pt.sqr(eta - 0.3).sum(axis=1))
    elif use == 2:
        m.Potential("p", pt.logsumexp(eta, axis=1) * -0.7 + pt.tanh(eta[:, 0]))
    else:
        m.StudentT("y", 4.0, mu=pt.dot(X, b) + 0.1 * eta[:, int(rng.integers(0, K))], sigma=s, observed=rng.normal(size=Nr))
    return m


@pytest.mark.parametrize("seed", range(80))
def test_random_matrix_products_keep_value_and_gradient(seed):
    m = _dot_model(seed)
    try:
        spec = lower_to_spec(m)
    except NotLowerable as e:
        assert "instruction" in str(e) or "longer" in str(e), str(e)
        pytest.skip(f"refused: {e}")
    rng = np.random.default_rng(7000 + seed)
    for scale in (0.3, 0.8):
        q = rng.normal(size=spec.n) * scale
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert np.isfinite(lp0)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (seed, scale, lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (seed, scale, float(np.max(np.abs(g - g0))))


def test_a_second_regression_likelihood_next_to_the_glm_node_is_written_out():
    """Two outcomes that share their coefficients: the first likelihood takes the model's ONE GLM node; the second one's `dot` (inner
    dimension 5) is written out inside its own element-wise factor instead of the refusal `more than one dense node`."""
    rng = np.random.default_rng(3)
    N, P = 40, 5
    X, X2 = rng.normal(size=(N, P)), rng.normal(size=(25, P))
    m = sg.StubModel()
    b = m.Normal("b", 0.0, 1.0, shape=(P,))
    s = m.HalfNormal("s", 1.0)
    m.Normal("y1", mu=pt.dot(sg.as_tensor(X), b), sigma=s, observed=rng.normal(size=N))
    m.Poisson("y2", pt.exp(pt.dot(sg.as_tensor(X2), b)), observed=rng.poisson(2.0, size=25).astype("float64"))
    spec = lower_to_spec(m)
    assert spec.glm_rows is not None and spec.glm_rows.X.shape == (N, P) and [f.size for f in spec.factors if f.name == "y2"] == [25]
    for scale in (0.2, 0.7):
        q = rng.normal(size=spec.n) * scale
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)) and np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0)))


# ---- covariance factors that are variables of the model: `pm.LKJCholeskyCov` of random size, concentration and standard-deviation prior, used
# through a non-centred product, as the `chol` of an observed `pm.MvNormal`, or both ----------------------------------------------------
def _lkj_model(seed):
    rng = np.random.default_rng(51000 + seed)
    n = int(rng.integers(2, 5))
    m = sg.StubModel()
    sd = [("Exponential", dict(lam=float(rng.uniform(0.5, 2.0)))), ("HalfNormal", dict(sigma=float(rng.uniform(0.5, 3.0)))), ("Gamma", dict(alpha=2.0, beta=1.5))][int(rng.integers(0, 3))]
    chol = m.LKJCholeskyCov("chol", n=n, eta=float([1.0, 1.5, 2.0, 4.0][int(rng.integers(0, 4))]), sd_dist=sd)
    use = int(rng.integers(0, 4))
    if use == 3:          # centred: the effects themselves under `MvNormal(mu, chol=chol)`, a row per group
        J = int(rng.integers(2, 6))
        mu = m.Normal("mu", 0.0, 2.0, shape=(n,))
        ab = m.MvNormal("ab", mu=mu, chol=chol, shape=(J, n))
        g = rng.integers(0, J, size=12)
        m.Normal("y3", mu=ab[:, 0][g] + ab[:, n - 1][g] * rng.normal(size=12), sigma=0.7, observed=rng.normal(size=12))
        return m
    if use in (0, 2):
        J = int(rng.integers(2, 6))
        z = m.Normal("z", 0.0, 1.0, shape=(n, J))
        ab = pt.dot(chol, z)
        g = rng.integers(0, J, size=12)
        x = rng.normal(size=12)
        m.Normal("y1", mu=ab[0][g] + ab[n - 1][g] * x, sigma=0.7, observed=rng.normal(size=12))
    if use in (1, 2):
        mu = m.Normal("mu", 0.0, 2.0, shape=(n,))
        rows = int(rng.integers(1, 7))
        Y = rng.normal(size=(rows, n)) if rows > 1 else rng.normal(size=(n,))
        m.MvNormal("y2", mu=mu * 0.5 if rng.uniform() < 0.5 else mu, chol=chol, observed=Y)
    return m


@pytest.mark.parametrize("seed", range(40))
def test_random_models_with_a_random_covariance_factor_keep_value_and_gradient(seed):
    m = _lkj_model(seed)
    spec = lower_to_spec(m)
    rng = np.random.default_rng(8000 + seed)
    for scale in (0.3, 0.7):
        q = rng.normal(size=spec.n) * scale
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert np.isfinite(lp0)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)), (seed, scale, lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0))), (seed, scale, float(np.max(np.abs(g - g0))))


def test_a_multivariate_normal_with_an_expression_for_its_mean_and_a_constant_covariance_is_written_out():
    """`y ~ MvNormal(a + b x, cov)` observed (a small Gaussian process with a fixed kernel and a parametric mean) and `beta ~
    MvNormal(m0 1, cov)` free: not the MvNormal node's (its mean is a constant) -- `solve_lower(cholesky(cov), value - mu)` with the
    constant factor against a vector of expressions is a short product with its inverse.  Against SciPy's multivariate normal."""
    from scipy import stats

    rng = np.random.default_rng(12)
    k = 5
    A = rng.normal(size=(k, k))
    S = A @ A.T / k + 0.5 * np.eye(k)
    xs, Y, y2 = np.linspace(-1, 1, k), rng.normal(size=(3, k)), rng.normal(size=k)
    m = sg.StubModel()
    a = m.Normal("a", 0.0, 2.0)
    b = m.Normal("b", 0.0, 2.0)
    m.MvNormal("y", mu=a + b * sg.as_tensor(xs), cov=S, observed=Y)
    m0 = m.Normal("m0", 0.0, 1.0)
    beta = m.MvNormal("beta", mu=m0 * sg.as_tensor(np.ones(k)), cov=S)
    m.Normal("y2", mu=beta, sigma=0.5, observed=y2)
    spec = lower_to_spec(m)
    assert spec.mvnormal is None and [f.size for f in spec.factors if f.name == "beta"] == [1]
    for scale in (0.3, 0.8):
        q = rng.normal(size=spec.n) * scale
        want = stats.norm(0, 2).logpdf(q[:2]).sum() + stats.multivariate_normal(np.zeros(k), S).logpdf(Y - (q[0] + q[1] * xs)).sum() + stats.norm(0, 1).logpdf(q[2])
        want += stats.multivariate_normal(q[2] * np.ones(k), S).logpdf(q[3:]) + stats.norm(q[3:], 0.5).logpdf(y2).sum()
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert abs(lp0 - want) <= 1e-10 * abs(want) and abs(lp - want) <= 1e-10 * abs(want)
        assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0)))


def test_the_centred_multivariate_hierarchy_has_scipys_multivariate_normal_for_its_effects():
    """`ab ~ MvNormal(mu, chol=chol, shape=(J, 2))` with `chol` from `LKJCholeskyCov`: J values of the density for a variable of 2 J
    elements, read through index vectors -- against SciPy's `multivariate_normal(mu, L L^T)` over the rows."""
    from scipy import stats

    from pymc_amd import model_spec as ms

    rng = np.random.default_rng(21)
    J, N = 6, 30
    cty, x, y = rng.integers(0, J, size=N), rng.normal(size=N), rng.normal(size=N)
    m = sg.StubModel()
    chol = m.LKJCholeskyCov("chol", n=2, eta=2.0, sd_dist=("Exponential", dict(lam=1.0)))
    mu = m.Normal("mu", 0.0, 3.0, shape=(2,))
    ab = m.MvNormal("ab", mu=mu, chol=chol, shape=(J, 2))
    m.Normal("y", mu=ab[:, 0][cty] + ab[:, 1][cty] * x, sigma=0.6, observed=y)
    spec = lower_to_spec(m)
    only = ms.ModelSpec(vars=spec.vars, data=spec.data, factors=[f for f in spec.factors if f.name.split(".")[0] == "ab"])
    assert [f.size for f in only.factors] == [J] and ms.engine_refusal(spec) is None
    for scale in (0.3, 0.8):
        q = rng.normal(size=spec.n) * scale
        v = q[:3]
        L = np.array([[np.exp(v[0]), 0.0], [v[1], np.exp(v[2])]])
        want = stats.multivariate_normal(q[3:5], L @ L.T).logpdf(q[5:].reshape(J, 2)).sum()
        assert abs(ref_models.evaluate(only, q)[0] - want) <= 1e-10 * max(1.0, abs(want))
        lp0, g0 = gt.joint_logp_grad(m, q)
        lp, g = ref_models.evaluate(spec, q)
        assert abs(lp - lp0) <= 1e-10 * max(1.0, abs(lp0)) and np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0)))


def test_a_matrix_valued_deterministic_keeps_its_shape_in_the_trace():
    """`pm.OrderedProbit(..., compute_p=True)` registers its [N, K] probabilities as `<name>_probs` (discrete.py:1403-1408): the trace holds
    (draws, N, K), every row summing to one."""
    from pymc_amd.backends import NDArray

    rng = np.random.default_rng(2)
    N = 6
    m = sg.StubModel()
    b = m.Normal("b", 0.0, 2.0)
    c = m.Normal("c", np.array([-1.0, 1.0]), 2.0, shape=(2,), transform="ordered")
    p = sg.reference()["OrderedProbit"].compute_p(b * sg.as_tensor(rng.normal(size=N)), c, 1.0)
    m.Deterministic("y_probs", p)
    m.Categorical("y", p=p, observed=rng.integers(0, 3, size=N).astype("float64"))
    spec = lower_to_spec(m)
    assert spec.deterministic_shapes == {"y_probs": (N, 3)}
    tr = NDArray(model=spec)
    tr.setup(3, 0)
    tr.record_batch(rng.normal(size=(3, spec.n)), None)
    assert tr.samples["y_probs"].shape == (3, N, 3) and np.allclose(tr.samples["y_probs"].sum(axis=-1), 1.0, atol=1e-14)
    assert tr.samples["c"].shape == (3, 2) and np.all(np.diff(tr.samples["c"], axis=1) > 0)
