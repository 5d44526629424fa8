"""Models whose matrix products become LINEAR PREDICTORS (dense node 5, include/nuts_mi355.h `nuts_lin`; tests/test_lin_lowering.py):
`pm.math.dot(X, beta)` over a tall constant design matrix, or over an inner dimension too long to write out, inside arguments the GLM
node's three likelihoods do not cover -- and `pt.sum(x)` over a long vector.  As in tests/more_models.py the graphs are what the
reference's own `logp` bodies build on the graph protocol of tests/stubgraph.py; committed: the graphs (tests/golden/lin_graphs.npz)
and torch autograd of them at seeded points (tests/golden/lin_graphs_golden.npz), written by tests/golden/make_lin_golden.py."""
import os

import numpy as np

import stubgraph as sg

pt = sg.pt
_rg = np.random.default_rng(20260930)

N_TS, P_TS, K_TS = 1200, 4, 3
X_TS = _rg.normal(size=(N_TS, P_TS))
_B = _rg.normal(size=(P_TS, K_TS))
Y_TS = np.array([_rg.choice(K_TS, p=np.exp(e - e.max()) / np.exp(e - e.max()).sum()) for e in X_TS @ _B + np.array([0.2, -0.3, 0.0])], dtype="float64")


def tall_softmax_regression():
    """`y ~ Categorical(p = softmax(X @ B + a))` over 1 200 rows: the product is not written out (tests/more_models.py does that for 90
    rows) -- its K columns are linear predictors sharing X, the intercepts stay gathers, the softmax row and `Categorical.logp`'s
    selection are the rows' program (discrete.py:1173-1205, math.py `softmax`)."""
    m = sg.StubModel()
    B = m.Normal("B", 0.0, 2.0, shape=(P_TS, K_TS))
    a = m.Normal("a", 0.0, 2.0, shape=(K_TS,))
    m.Categorical("y", p=pt.softmax(pt.dot(sg.as_tensor(X_TS), B) + a[None, :], axis=-1), observed=Y_TS)
    return m


N_TR, P_TR = 1500, 5
X_TR = _rg.normal(size=(N_TR, P_TR))
Y_TR = X_TR @ np.array([0.8, -0.5, 0.0, 1.1, 0.3]) + 0.4 * _rg.standard_t(3.0, size=N_TR)


def tall_robust_regression():
    """`pm.StudentT(nu, mu = pm.math.dot(X, b), sigma = s)` with a random nu (continuous.py:1935-1950): one predictor as the location
    of a density lowered op by op."""
    m = sg.StubModel()
    b = m.Normal("b", 0.0, 2.0, shape=(P_TR,))
    s = m.HalfNormal("s", 1.0)
    nu = m.Gamma("nu", 2.0, 0.1)
    m.StudentT("y", nu, mu=pt.dot(sg.as_tensor(X_TR), b), sigma=s, observed=Y_TR)
    return m


N_NB, P_NB = 1100, 3
X_NB = _rg.normal(size=(N_NB, P_NB)) * 0.5
Y_NB = _rg.negative_binomial(3.0, 3.0 / (3.0 + np.exp(1.0 + X_NB @ np.array([0.6, -0.4, 0.2])))).astype("float64")


def negative_binomial_regression():
    """Counts with over-dispersion: `pm.NegativeBinomial(mu = exp(a + dot(X, b)), alpha)` (discrete.py:668-745) -- the log link of the
    GLM node's Poisson family under a likelihood it does not have."""
    m = sg.StubModel()
    a = m.Normal("a", 0.0, 3.0)
    b = m.Normal("b", 0.0, 1.0, shape=(P_NB,))
    alpha = m.Exponential("alpha", 0.5)
    m.NegativeBinomial("y", mu=pt.exp(a + pt.dot(sg.as_tensor(X_NB), b)), alpha=alpha, observed=Y_NB)
    return m


N_WC, P_WC = 200, 40
X_WC = _rg.normal(size=(N_WC, P_WC)) / np.sqrt(P_WC)
Y_WC = X_WC @ (_rg.normal(size=P_WC) * 0.7 + 0.2) + 0.3 * _rg.standard_cauchy(size=N_WC)


def wide_noncentred_cauchy():
    """Forty covariates (an inner dimension the lowering does not write out) with non-centred coefficients under a Cauchy likelihood:
    `pm.Cauchy(alpha = dot(X, mu + tau * z), beta = 0.5)` -- the coefficients are an EXPRESSION of the variables, a derived vector the
    predictor reads and seeds."""
    m = sg.StubModel()
    mu = m.Normal("mu", 0.0, 1.0)
    tau = m.HalfNormal("tau", 1.0)
    z = m.Normal("z", 0.0, 1.0, shape=(P_WC,))
    m.Cauchy("y", alpha=pt.dot(sg.as_tensor(X_WC), mu + tau * z), beta=0.5, observed=Y_WC)
    return m


L_SZ, M_SZ = 1500, 300
Y_SZ = 0.4 + _rg.normal(size=M_SZ)


def long_sums():
    """Reductions over a long axis inside arguments: observations centred on `c + sum(x) / L` (the mean of a 1 500-vector, broadcast
    against every observation) and a soft sum-to-zero constraint `Normal(sum(x) | 0, 0.01 L)` -- one-row predictors over the same
    variable."""
    m = sg.StubModel()
    c = m.Normal("c", 0.0, 3.0)
    x = m.Normal("x", 0.0, 1.0, shape=(L_SZ,))
    m.Normal("y", mu=c + pt.sum(x) / float(L_SZ), sigma=1.0, observed=Y_SZ)
    m.Normal("sum0", mu=pt.sum(x), sigma=0.01 * L_SZ, observed=np.array(0.0))
    return m


T_RW = 240
_h = 1.2 + 0.15 * np.cumsum(_rg.normal(size=T_RW))
Y_RW = _rg.poisson(np.exp(_h)).astype("float64")


def noncentred_random_walk_rate():
    """A log-rate that wanders, NON-CENTRED: `eps ~ Normal(0, 1)[240]`, `y_t ~ Poisson(exp(h0 + s * cumsum(eps)_t))` -- the running sum
    of a long vector (`CumOp`; what `pm.GaussianRandomWalk` spares the user is exactly this parameterisation when the data are weak).
    The lowering reads `cumsum` over more than 32 elements as the product with the lower-triangular matrix of ones: a linear predictor
    with T rows and T columns (round 6)."""
    m = sg.StubModel()
    h0 = m.Normal("h0", 1.0, 2.0)
    s = m.HalfNormal("s", 0.3)
    eps = m.Normal("eps", 0.0, 1.0, shape=(T_RW,))
    m.Poisson("y", mu=pt.exp(h0 + s * pt.cumsum(eps)), observed=Y_RW)
    return m


def _lattice(rows, cols):
    n = rows * cols
    W = np.zeros((n, n), dtype=np.int64)
    for i in range(rows):
        for j in range(cols):
            k = i * cols + j
            if j + 1 < cols:
                W[k, k + 1] = W[k + 1, k] = 1
            if i + 1 < rows:
                W[k, k + cols] = W[k + cols, k] = 1
    return W


W_ICAR = _lattice(7, 8)
E_ICAR = 20.0 + 10.0 * np.cos(np.arange(56) * 0.7)
Y_ICAR = np.floor(E_ICAR * np.exp(0.3 * np.sin(np.arange(56) * 0.5)))                  # (no random draws)


def icar_over_fifty_six_areas():
    """`pm.ICAR` (multivariate.py:2315-2447) over a 7 x 8 lattice: 97 edges and 56 areas -- the sum of the squared differences over the
    edge list and the sum of the areas' effects are sums over LONG vectors (tests/more_models.py has the same model over 20 areas, written
    out): linear predictors with a row of ones."""
    m = sg.StubModel()
    sigma = m.Exponential("sigma", 1.0)
    b0 = m.Normal("b0", 0.0, 1.0)
    phi = m.ICAR("phi", W=W_ICAR, sigma=sigma)
    m.Poisson("y", mu=pt.exp(sg.as_tensor(np.log(E_ICAR)) + b0 + phi), observed=Y_ICAR)
    return m


_rg2 = np.random.default_rng(20260931)      # (a generator of its own: the arrays above keep their values)
REG_RW = (np.arange(T_RW) // 80).astype(np.int64)
Y_RW3 = _rg2.poisson(np.exp(1.2 + 0.1 * np.cumsum(_rg2.normal(size=T_RW)))).astype("float64")


def regime_scaled_random_walk():
    """The non-centred random walk with an innovation scale PER REGIME: `cumsum(s[regime] * eps)` -- the coefficient vector of the
    predictor is a derived vector that reads `s` through an index vector (its seed arrives from the predictor's backward pass: the
    ordering the 56-area ICAR got wrong before the fix, here with an owning variable)."""
    m = sg.StubModel()
    h0 = m.Normal("h0", 1.0, 2.0)
    s = m.HalfNormal("s", 0.3, shape=(3,))
    eps = m.Normal("eps", 0.0, 1.0, shape=(T_RW,))
    m.Poisson("y", mu=pt.exp(h0 + pt.cumsum(s[REG_RW] * eps)), observed=Y_RW3)
    return m


G_VS = 12
GI_VS = ((np.arange(N_TR) * 7) % G_VS).astype(np.int64)
Y_VS = Y_TR + 0.3 * np.sin(GI_VS)
X2_TR = _rg2.normal(size=(N_TR, 3))


def varying_intercepts_and_scales_under_a_predictor():
    """`StudentT(nu, mu = dot(X, b) + a[g], sigma = s[g])` over 1 500 rows: a linear predictor, two gathers and a scalar that broadcasts,
    all operands of one factor's program."""
    m = sg.StubModel()
    b = m.Normal("b", 0.0, 2.0, shape=(P_TR,))
    a = m.Normal("a", 0.0, 1.0, shape=(G_VS,))
    s = m.HalfNormal("s", 1.0, shape=(G_VS,))
    nu = m.Gamma("nu", 2.0, 0.1)
    m.StudentT("y", nu, mu=pt.dot(sg.as_tensor(X_TR), b) + a[GI_VS], sigma=s[GI_VS], observed=Y_VS)
    return m


def two_design_matrices():
    """`Laplace(mu = dot(X1, b1) + tanh(dot(X2, b2)), b = s)`: two predictors over two different constant matrices in one argument."""
    m = sg.StubModel()
    b1 = m.Normal("b1", 0.0, 2.0, shape=(P_TR,))
    b2 = m.Normal("b2", 0.0, 2.0, shape=(3,))
    s = m.HalfNormal("s", 1.0)
    m.Laplace("y", mu=pt.dot(sg.as_tensor(X_TR), b1) + pt.tanh(pt.dot(sg.as_tensor(X2_TR), b2)), b=s, observed=Y_TR)
    return m


MODELS = {
    "regime_scaled_random_walk": regime_scaled_random_walk,
    "varying_intercepts_and_scales_under_a_predictor": varying_intercepts_and_scales_under_a_predictor,
    "two_design_matrices": two_design_matrices,
    "icar_over_fifty_six_areas": icar_over_fifty_six_areas,
    "noncentred_random_walk_rate": noncentred_random_walk_rate,
    "tall_softmax_regression": tall_softmax_regression,
    "tall_robust_regression": tall_robust_regression,
    "negative_binomial_regression": negative_binomial_regression,
    "wide_noncentred_cauchy": wide_noncentred_cauchy,
    "long_sums": long_sums,
}
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "lin_graphs.npz")
GOLDEN = os.path.join(HERE, "golden", "lin_graphs_golden.npz")
