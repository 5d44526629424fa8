"""More models for the lowering (tests/test_more_lowering.py).  Time series (distributions/timeseries.py): the stochastic-volatility model
of the reference's own example gallery, autoregressions, a random-walk rate under counts -- the densities are the reference's
`random_walk_logp` / `logprob_cumsum` / `logprob_join` / `ar_logp` bodies and its distributions' `logp`, executed on tests/stubgraph.py.
Zero-sum effects (`pm.ZeroSumNormal`, multivariate.py:2654-2807, under `ZeroSumTransform`, transforms.py:644-696): group effects that sum
to zero next to an intercept.  Matrix products outside the dense nodes: a softmax regression with a [P, K] coefficient matrix, a robust
regression whose location is `pm.math.dot(X, beta)`.

Round 5 validated these specs on the host only; since round 6 `tests/test_more_lowering.py` has a `-m gpu` half over all of them (device
log-density / gradient against the committed autograd goldens, NUTS integers against the oracle sampler).  They stay out of
`lowering_models.GENERAL` and of tests/golden/lowered_spec_digests.json (their own fixture: tests/golden/more_graphs.npz)."""
import os

import numpy as np

import stubgraph as sg

pt = sg.pt
_rg = np.random.default_rng(20240925)
T_SV = 100
_vol_true = np.cumsum(_rg.normal(size=T_SV) * 0.12) - 3.5
RETURNS = _rg.standard_t(6.0, size=T_SV) * np.exp(_vol_true)
T_AR = 60
_x = np.zeros(T_AR)
for _t in range(2, T_AR):
    _x[_t] = 0.3 + 0.55 * _x[_t - 1] - 0.25 * _x[_t - 2] + 0.4 * _rg.normal()
Y_AR = _x + 0.3 * _rg.normal(size=T_AR)
T_LL = 48
COUNTS = _rg.poisson(np.exp(1.0 + np.cumsum(_rg.normal(size=T_LL) * 0.15))).astype("float64")


def stochastic_volatility():
    """The reference gallery's stochastic-volatility model: `step_size ~ Exponential(10)`; `volatility ~ GaussianRandomWalk(sigma =
    step_size, init_dist = Normal.dist(0, 100))`; `nu ~ Exponential(0.1)`; `returns ~ StudentT(nu, lam = exp(-2 volatility))`."""
    m = sg.StubModel()
    step = m.Exponential("step_size", 10.0)
    vol = m.GaussianRandomWalk("volatility", sigma=step, init_dist=("Normal", dict(mu=0.0, sigma=100.0)), shape=(T_SV,))
    nu = m.Exponential("nu", 0.1)
    m.StudentT("returns", nu, lam=pt.exp(-2.0 * vol), observed=RETURNS)
    return m


def ar2_with_constant():
    """`pm.AR(rho = [c, r1, r2], constant = True)` (timeseries.py:420-644) as a latent state under noisy observations."""
    m = sg.StubModel()
    rho = m.Normal("rho", 0.0, 0.5, shape=(3,))
    s = m.HalfNormal("s", 1.0)
    x = m.AR("x", rho, sigma=s, init_dist=("Normal", dict(mu=0.0, sigma=2.0)), constant=True, shape=(T_AR,))
    tau = m.HalfNormal("tau", 0.5)
    m.Normal("y", mu=x, sigma=tau, observed=Y_AR)
    return m


def ar1_latent():
    """AR(1) without a constant term: one coefficient under a Uniform(-1, 1) prior (interval transform)."""
    m = sg.StubModel()
    rho = m.Uniform("rho", -1.0, 1.0, shape=(1,))
    x = m.AR("x", rho, sigma=0.4, init_dist=("Normal", dict(mu=0.0, sigma=1.0)), constant=False, shape=(T_AR,))
    m.Normal("y", mu=x, sigma=0.3, observed=Y_AR)
    return m


def random_walk_rate_under_counts():
    """A local-level model of counts: the log-rate is a Gaussian random walk with a drift, `counts ~ Poisson(exp(level))`."""
    m = sg.StubModel()
    drift = m.Normal("drift", 0.0, 0.1)
    step = m.HalfNormal("step", 0.3)
    level = m.GaussianRandomWalk("level", mu=drift, sigma=step, init_dist=("Normal", dict(mu=1.0, sigma=2.0)), shape=(T_LL,))
    m.Poisson("counts", pt.exp(level), observed=COUNTS)
    return m


K_ZS, N_ZS = 9, 72
G_ZS = _rg.integers(0, K_ZS, size=N_ZS)
_eff = _rg.normal(size=K_ZS) * 0.8
Y_ZS = 1.5 + (_eff - _eff.mean())[G_ZS] + 0.5 * _rg.normal(size=N_ZS)
Y_ZS2 = _rg.poisson(np.exp(0.8 + 0.4 * (_eff - _eff.mean()))).astype("float64")


def zero_sum_group_effects():
    """An intercept and group effects that sum to zero (the identifiable form of a one-way layout): `z ~ ZeroSumNormal(sigma, shape=K)`,
    `y ~ Normal(a + z[group], s)`.  K - 1 free values; the K-th effect balances them (`ZeroSumTransform.extend_axis`), so `z[group]` is a
    gather that spans both pieces of a concatenation."""
    m = sg.StubModel()
    a = m.Normal("a", 0.0, 5.0)
    tau = m.HalfNormal("tau", 1.0)
    z = m.ZeroSumNormal("z", sigma=tau, shape=(K_ZS,))
    s = m.HalfNormal("s", 1.0)
    m.Normal("y", mu=a + z[G_ZS], sigma=s, observed=Y_ZS)
    return m


def zero_sum_log_rates():
    """The zero-sum vector used element-wise: one count per group, `counts ~ Poisson(exp(a + z))`."""
    m = sg.StubModel()
    a = m.Normal("a", 0.0, 2.0)
    z = m.ZeroSumNormal("z", sigma=0.7, shape=(K_ZS,))
    m.Poisson("counts", pt.exp(a + z), observed=Y_ZS2)
    return m


N_SM, P_SM, K_SM = 90, 4, 3
X_SM = _rg.normal(size=(N_SM, P_SM))
_B = _rg.normal(size=(P_SM, K_SM)) * 1.2
Y_SM = np.array([_rg.choice(K_SM, p=np.exp(e - e.max()) / np.exp(e - e.max()).sum()) for e in X_SM @ _B + np.array([0.3, -0.2, 0.0])], dtype="float64")
Y_RB = X_SM @ np.array([0.8, -0.5, 0.0, 1.1]) + 0.4 * _rg.standard_t(3.0, size=N_SM)


def softmax_regression():
    """Multinomial logistic regression, the way it is usually written: `B` a [P, K] matrix of coefficients, `a` K intercepts,
    `y ~ Categorical(p = softmax(X @ B + a))`.  The matrix product is no dense node's: over a short inner dimension it is written out
    (element (n, k) = sum_p X[n, p] B[p, k]: the column of X a data vector, the element of B a gather), then the softmax row and
    `Categorical.logp`'s selection of the observed category."""
    m = sg.StubModel()
    B = m.Normal("B", 0.0, 2.0, shape=(P_SM, K_SM))
    a = m.Normal("a", 0.0, 2.0, shape=(K_SM,))
    m.Categorical("y", p=pt.softmax(pt.dot(sg.as_tensor(X_SM), B) + a[None, :], axis=-1), observed=Y_SM)
    return m


def robust_regression_with_dot():
    """`pm.math.dot(X, beta)` as the location of a StudentT with a random nu: not the GLM node's families, so the product is written out
    inside the likelihood's program."""
    m = sg.StubModel()
    b = m.Normal("b", 0.0, 2.0, shape=(P_SM,))
    s = m.HalfNormal("s", 1.0)
    nu = m.Gamma("nu", 2.0, 0.1)
    m.StudentT("y", nu, mu=pt.dot(sg.as_tensor(X_SM), b), sigma=s, observed=Y_RB)
    return m


N_OP = 54
X_OP = _rg.normal(size=N_OP)
Y_OP = np.digitize(1.1 * X_OP + _rg.normal(size=N_OP), [-0.8, 0.9]).astype("float64")
Y_OP4 = np.digitize(0.9 * X_OP - 0.2 + 0.8 * np.cos(np.arange(N_OP) * 1.7), [-0.9, 0.1, 1.0]).astype("float64")   # (no new random draws: later data stay put)
N_ZI = 60
_zi_keep = _rg.uniform(size=N_ZI) < 0.65
Y_ZIB = (_rg.binomial(12, 0.3, size=N_ZI) * _zi_keep).astype("float64")
Y_ZINB = (_rg.negative_binomial(2.0, 2.0 / (2.0 + 3.0), size=N_ZI) * _zi_keep).astype("float64")


def ordered_probit_three_levels():
    """`pm.OrderedProbit` (discrete.py:1329-1432) with ordered cutpoints: three levels -- every level's probability is a
    `log_diff_normal_cdf` / `normal_lcdf` body of some twenty-five instructions and `Categorical.logp` checks all of them: 100 of a
    factor's 128 instructions."""
    m = sg.StubModel()
    b = m.Normal("b", 0.0, 2.0)
    c = m.Normal("c", np.array([-1.0, 1.0]), 2.0, shape=(2,), transform="ordered")
    m.OrderedProbit("y", eta=b * sg.as_tensor(X_OP), cutpoints=c, observed=Y_OP)
    return m


def ordered_probit_four_levels():
    """Four levels: density and checks no longer fit ONE program of 128 instructions -- the density keeps the factor, each of
    `Categorical.logp`'s three parameter checks (`0 <= p`, `p <= 1`, `isclose(sum(p), 1)`) becomes a factor `check(0, cond)` of its own."""
    m = sg.StubModel()
    b = m.Normal("b", 0.0, 2.0)
    c = m.Normal("c", np.array([-1.0, 0.0, 1.0]), 2.0, shape=(3,), transform="ordered")
    m.OrderedProbit("y", eta=b * sg.as_tensor(X_OP), cutpoints=c, observed=Y_OP4)
    return m


def zero_inflated_binomial_and_negative_binomial():
    """`pm.ZeroInflatedBinomial`, `pm.ZeroInflatedNegativeBinomial` (mixture.py:641-800) sharing one inflation probability."""
    m = sg.StubModel()
    psi = m.Beta("psi", 2.0, 2.0)
    p = m.Beta("p", 2.0, 2.0)
    m.ZeroInflatedBinomial("yb", psi, 12, p, observed=Y_ZIB)
    mu = m.Gamma("mu", 2.0, 0.5)
    al = m.Exponential("al", 0.5)
    m.ZeroInflatedNegativeBinomial("ynb", psi, mu, al, observed=Y_ZINB)
    return m


Y_TR1 = _rg.uniform(0.3, 2.4, size=20)
Y_TR2 = _rg.uniform(1.1, 6.0, size=15)
Y_TR3 = _rg.uniform(-3.0, 0.9, size=12)


def truncated_likelihoods():
    """`pm.Truncated` (distributions/truncated.py:418-458) over three base families, truncated on both sides, below only, above only: the
    base density less the log of the mass between the bounds -- `logdiffexp` of the base's `logcdf` at the two bounds, its `logccdf` at
    the lower one, its `logcdf` at the upper one -- behind the support switches."""
    m = sg.StubModel()
    lam = m.HalfNormal("lam", 2.0)
    m.Truncated("e", ("Exponential", dict(lam=lam)), 0.2, 2.5, observed=Y_TR1)
    mu = m.Normal("mu", 0.0, 2.0)
    s = m.HalfNormal("s", 2.0)
    m.Truncated("lo", ("Laplace", dict(mu=mu, b=s)), 1.0, None, observed=Y_TR2)
    m.Truncated("up", ("Logistic", dict(mu=mu, s=s)), None, 1.0, observed=Y_TR3)
    return m


J_VS, N_VS = 7, 56
CTY_VS = np.arange(N_VS) % J_VS
FLOOR_VS = ((np.arange(N_VS) * 7) % 3 == 0).astype("float64")
Y_VS = 1.4 + 0.3 * np.sin(CTY_VS * 1.3) - (0.6 + 0.2 * np.cos(CTY_VS * 0.7)) * FLOOR_VS + 0.35 * np.sin(np.arange(N_VS) * 2.1)   # (no random draws)


def varying_slopes_lkj():
    """Varying intercepts and slopes with an LKJ prior, non-centred -- the multilevel model of the reference's gallery: `chol, _, _ =
    pm.LKJCholeskyCov("chol", n=2, eta=2, sd_dist=pm.Exponential.dist(1))`, `z ~ Normal(0, 1, shape=(2, J))`, `ab = pm.math.dot(chol, z)`,
    `y ~ Normal((mu[0] + ab[0][county]) + (mu[1] + ab[1][county]) * floor, s)`.  The packed factor's density (multivariate.py:1271-1310) is
    cumulative sums, `inc_subtensor`s and integer-array indices over three values; `expand_packed_triangular` a `set_subtensor` at
    `np.tril_indices`; the product with z has an inner dimension of two."""
    m = sg.StubModel()
    chol = m.LKJCholeskyCov("chol", n=2, eta=2.0, sd_dist=("Exponential", dict(lam=1.0)))
    z = m.Normal("z", 0.0, 1.0, shape=(2, J_VS))
    mu = m.Normal("mu", 0.0, 5.0, shape=(2,))
    ab = pt.dot(chol, z)
    s = m.HalfNormal("s", 1.0)
    m.Normal("y", mu=(mu[0] + ab[0][CTY_VS]) + (mu[1] + ab[1][CTY_VS]) * FLOOR_VS, sigma=s, observed=Y_VS)
    return m


def three_correlated_effects_lkj():
    """n = 3, eta = 1 (the other branch of `_lkj_normalizing_constant`), HalfNormal standard deviations."""
    m = sg.StubModel()
    chol = m.LKJCholeskyCov("chol", n=3, eta=1.0, sd_dist=("HalfNormal", dict(sigma=2.0)))
    z = m.Normal("z", 0.0, 1.0, shape=(3, J_VS))
    ab = pt.dot(chol, z)
    m.Normal("y", mu=ab[0][CTY_VS] + ab[1][CTY_VS] * FLOOR_VS + 0.5 * ab[2][CTY_VS], sigma=0.6, observed=Y_VS)
    return m


Y_MV2 = np.stack([0.8 * np.sin(np.arange(24) * 0.9) + 0.5, 0.6 * np.sin(np.arange(24) * 0.9 + 0.4) - 0.3 + 0.2 * np.cos(np.arange(24) * 2.3)], axis=1)
Y_MV3 = np.concatenate([Y_MV2, (0.5 * Y_MV2[:, :1] - 0.7 * Y_MV2[:, 1:] + 0.3 * np.cos(np.arange(24) * 1.1)[:, None])], axis=1)


def multivariate_outcomes_lkj():
    """`y ~ MvNormal(mu, chol=chol)` with `chol` from `pm.LKJCholeskyCov`: a covariance factor that is a VARIABLE of the model.
    `quaddist_matrix(chol=...)` forms `chol @ chol.mT` and tags chol lower-triangular, `quaddist_chol` takes `cholesky` of that product
    (-> chol: the rewrite the tag exists for), `solve_lower(chol, y - mu, b_ndim=1)` is forward substitution over two columns, the log-
    determinant the log of the diagonal (multivariate.py:128-185)."""
    m = sg.StubModel()
    chol = m.LKJCholeskyCov("chol", n=2, eta=2.0, sd_dist=("Exponential", dict(lam=1.0)))
    mu = m.Normal("mu", 0.0, 3.0, shape=(2,))
    m.MvNormal("y", mu=mu, chol=chol, observed=Y_MV2)
    return m


def three_outcomes_lkj():
    m = sg.StubModel()
    chol = m.LKJCholeskyCov("chol", n=3, eta=1.5, sd_dist=("HalfNormal", dict(sigma=2.0)))
    mu = m.Normal("mu", 0.0, 3.0, shape=(3,))
    m.MvNormal("y", mu=mu, chol=chol, observed=Y_MV3)
    return m


def correlated_outcomes_with_a_correlation_parameter():
    """`y ~ MvNormal(mu, cov=Sigma)` with Sigma = [[s0^2, rho s0 s1], [rho s0 s1, s1^2]] assembled from the model's own variables
    (`pt.stack`): the textbook bivariate normal with an explicit correlation.  `quaddist_matrix(cov=...)` hands the matrix over as it
    is, `quaddist_chol` takes `nan_lower_cholesky` of a matrix that is NOT a tagged product (multivariate.py:120-185): the factor of a
    small covariance matrix of expressions is written out element by element (`lowering.MAX_CHOLESKY`)."""
    m = sg.StubModel()
    s = m.HalfNormal("s", 2.0, shape=(2,))
    rho = m.Uniform("rho", -1.0, 1.0)
    mu = m.Normal("mu", 0.0, 3.0, shape=(2,))
    c = rho * s[0] * s[1]
    m.MvNormal("y", mu=mu, cov=pt.stack([pt.stack([s[0] ** 2, c]), pt.stack([c, s[1] ** 2])]), observed=Y_MV2)
    return m


def three_outcomes_with_a_banded_precision_matrix():
    """`y ~ MvNormal(mu, tau=T)` with T a tridiagonal precision matrix assembled from the model's variables (neighbouring outcomes
    conditionally dependent, the first and the last conditionally independent): `quaddist_matrix(tau=...)` takes `matrix_inverse(tau)`
    (multivariate.py:137-141) and `quaddist_chol` the Cholesky factor of that -- the 3 x 3 inverse written out as adjugate / determinant,
    then the factor element by element."""
    m = sg.StubModel()
    t = m.Gamma("t", 2.0, 1.0, shape=(3,))
    r = m.Uniform("r", -0.5, 0.5, shape=(2,))
    mu = m.Normal("mu", 0.0, 3.0, shape=(3,))
    z, a, b = sg.as_tensor(0.0), r[0] * pt.sqrt(t[0] * t[1]), r[1] * pt.sqrt(t[1] * t[2])
    m.MvNormal("y", mu=mu, tau=pt.stack([pt.stack([t[0], a, z]), pt.stack([a, t[1], b]), pt.stack([z, b, t[2]])]), observed=Y_MV3)
    return m


def heavy_tailed_correlated_outcomes():
    """`y ~ MvStudentT(nu, mu, chol=chol)` (multivariate.py:398-516) with the degrees of freedom, the location and an `LKJCholeskyCov`
    factor all variables: the same `quaddist_chol` as the MvNormal's, under `log1p(quaddist / nu)` and three `gammaln`s."""
    m = sg.StubModel()
    chol = m.LKJCholeskyCov("chol", n=2, eta=2.0, sd_dist=("Exponential", dict(lam=1.0)))
    nu = m.Gamma("nu", 2.0, 0.1)
    mu = m.Normal("mu", 0.0, 3.0, shape=(2,))
    m.MvStudentT("y", nu=nu, mu=mu, chol=chol, observed=Y_MV2)
    return m


Y_SKT = np.sin(np.arange(40) * 1.7) * 1.5 + 0.4 * np.cos(np.arange(40) * 0.3) + 0.8          # (no random draws)
C_DW = ((np.arange(40) * 7) % 6).astype("float64")


def skewed_measurements_and_discrete_lifetimes():
    """Two likelihoods the vocabulary lacked: the Jones-Faddy `pm.SkewStudentT(a, b, mu, sigma)` (continuous.py:2001-2078: `betaln` of
    two variables) and `pm.DiscreteWeibull(q, beta)` (discrete.py:430-510: a difference of two powers of powers) -- every parameter a
    variable of the model, lowered op by op."""
    m = sg.StubModel()
    a = m.Gamma("a", 3.0, 1.0)
    b = m.Gamma("b", 3.0, 1.0)
    mu = m.Normal("mu", 0.0, 3.0)
    s = m.HalfNormal("s", 2.0)
    m.SkewStudentT("y", a=a, b=b, mu=mu, sigma=s, observed=Y_SKT)
    q = m.Beta("q", 2.0, 2.0)
    beta = m.Gamma("beta", 2.0, 1.0)
    m.DiscreteWeibull("c", q=q, beta=beta, observed=C_DW)
    return m


N_HU = 50
X_HU = np.sin(np.arange(N_HU) * 0.9)
_Z_HU = ((np.arange(N_HU) * 5) % 4 != 0).astype("float64")
Y_HU_G = _Z_HU * (0.5 + np.abs(np.sin(np.arange(N_HU) * 1.3)) * 2.0)                      # (no random draws; a quarter of the amounts are zero)
Y_HU_L = np.roll(_Z_HU, 1) * np.exp(0.3 + 0.8 * np.cos(np.arange(N_HU) * 0.7))


def hurdle_models_of_positive_amounts():
    """`pm.HurdleGamma` and `pm.HurdleLogNormal` (mixture.py:805-870, 981-1090): zeros from a process of their own, positive amounts
    from a Gamma regression (mean `exp(a + b x)`) / a LogNormal whose hurdle probability depends on x -- the reference's
    `marginal_hurdle_logprob` with its safe value under the switch."""
    m = sg.StubModel()
    psi = m.Beta("psi", 2.0, 2.0)
    a = m.Normal("a", 0.0, 1.0)
    b = m.Normal("b", 0.0, 1.0)
    k = m.Gamma("k", 2.0, 1.0)
    m.HurdleGamma("y", psi=psi, alpha=k, beta=k / pt.exp(a + b * sg.as_tensor(X_HU)), observed=Y_HU_G)
    s = m.HalfNormal("s", 1.0)
    m.HurdleLogNormal("w", psi=pt.sigmoid(0.5 * a + b * sg.as_tensor(X_HU)), mu=b, sigma=s, observed=Y_HU_L)
    return m


def lattice_adjacency(rows, cols):
    """Rook adjacency of a rows x cols lattice: the symmetric 0 / 1 matrix `pm.ICAR` takes."""
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


W_CAR = lattice_adjacency(4, 5)
E_CAR = 20.0 + 10.0 * np.cos(np.arange(20) * 0.7)
Y_CAR = np.floor(E_CAR * np.exp(0.3 * np.sin(np.arange(20) * 0.5)))                    # (no random draws)


def disease_counts_over_a_lattice_of_areas():
    """Counts per area with an intrinsic conditional autoregression over the areas' adjacency (`pm.ICAR`, multivariate.py:2315-2447;
    the spatial part of the Besag-York-Mollie model of its docstring): `ICAR.logp` turns the constant adjacency matrix into an edge
    list -- `pt.eq(pt.tril(W), 1).nonzero()`, folded here -- sums the squared differences over the 31 edges and adds a soft
    sum-to-zero term over the 20 areas."""
    m = sg.StubModel()
    sigma = m.Exponential("sigma", 1.0)
    b0 = m.Normal("b0", 0.0, 1.0)
    phi = m.ICAR("phi", W=W_CAR, sigma=sigma)
    m.Poisson("y", mu=pt.exp(sg.as_tensor(np.log(E_CAR)) + b0 + phi), observed=Y_CAR)
    return m


X_GP4 = np.array([0.0, 0.7, 1.5, 2.6])
D2_GP4 = (X_GP4[:, None] - X_GP4[None, :]) ** 2
Y_GP4 = np.stack([np.sin(X_GP4 * 1.3 + 0.35 * r) * (1.0 + 0.1 * (r % 3)) + 0.12 * np.cos(np.arange(4) * 2.3 + r) for r in range(15)])   # (no random draws)


def replicated_curves_under_a_squared_exponential_kernel():
    """Fifteen curves observed at the same four inputs, `y_r ~ MvNormal(0, K)`, K = eta^2 exp(-d^2 / (2 ell^2)) + sigma^2 I with the
    amplitude, the length scale and the noise variables of the model: a Gaussian process's marginal likelihood over a handful of
    inputs (`pm.gp.Marginal` builds exactly this `MvNormal(cov=K)`), the 4 x 4 factor written out."""
    m = sg.StubModel()
    eta = m.HalfNormal("eta", 2.0)
    ell = m.Gamma("ell", 2.0, 2.0)
    sigma = m.HalfNormal("sigma", 1.0)
    K = eta ** 2 * pt.exp(sg.as_tensor(-0.5 * D2_GP4) / ell ** 2) + sigma ** 2 * sg.as_tensor(np.eye(4))
    m.MvNormal("y", mu=sg.as_tensor(np.zeros(4)), cov=K, observed=Y_GP4)
    return m


COUNTS_DM = np.array([[2, 6, 8, 4], [1, 9, 7, 3], [4, 4, 6, 6], [0, 7, 10, 3], [3, 5, 9, 3], [2, 8, 5, 5], [1, 4, 12, 3], [5, 6, 6, 3], [2, 7, 7, 4]], dtype="float64")


def over_dispersed_counts():
    """`counts ~ DirichletMultinomial(n, a = frac * conc)` (multivariate.py:690-790) with `frac ~ Dirichlet(1)`, `conc ~ LogNormal(1, 1)`: the
    reference's own docstring example.  K gammaln terms per row, reduced over the short axis."""
    m = sg.StubModel()
    frac = m.Dirichlet("frac", np.ones(4))
    conc = m.LogNormal("conc", 1.0, 1.0)
    m.DirichletMultinomial("counts", n=20, a=frac * conc, observed=COUNTS_DM)
    return m


T_EM, DT_EM = 64, 0.1
Y_EM = np.tanh(np.sin(np.arange(T_EM) * 0.35) * 2.0) + 0.15 * np.cos(np.arange(T_EM) * 2.9)


def double_well_sde():
    """`pm.EulerMaruyama` (timeseries.py:861-1003): dx = a (x - x^3) dt + s dW, the user's `sde_fn` called by the reference's
    `eulermaruyama_logp` -- x[t] ~ Normal(x[t-1] + dt f(x[t-1]), sqrt(dt) g) over slices of the path --, under noisy observations."""
    m = sg.StubModel()
    a = m.HalfNormal("a", 2.0)
    s = m.HalfNormal("s", 1.0)
    x = m.EulerMaruyama("x", DT_EM, lambda x_, a_, s_: (a_ * (x_ - x_ ** 3), s_), (a, s), init_dist=("Normal", dict(mu=0.0, sigma=2.0)), shape=(T_EM,))
    m.Normal("y", mu=x, sigma=0.2, observed=Y_EM)
    return m


N_NN = 80
X_NN = np.stack([np.cos(np.arange(N_NN) * 0.41) * (1.0 + 0.3 * (np.arange(N_NN) % 2)), np.sin(np.arange(N_NN) * 0.41) + 0.5 * (np.arange(N_NN) % 2)], axis=1)
Y_NN = (np.arange(N_NN) % 2).astype("float64")


def bayesian_neural_network():
    """The gallery's small network: two hidden layers of five tanh units, `out ~ Bernoulli(p = sigmoid(dot(tanh(dot(tanh(dot(X, W1)), W2)),
    w3)))`.  Three matrix products over short inner dimensions (2, 5, 5), none a dense node's: one program of 92 instructions."""
    m = sg.StubModel()
    W1 = m.Normal("w_in_1", 0.0, 1.0, shape=(2, 5))
    W2 = m.Normal("w_1_2", 0.0, 1.0, shape=(5, 5))
    w3 = m.Normal("w_2_out", 0.0, 1.0, shape=(5,))
    act = pt.tanh(pt.dot(pt.tanh(pt.dot(sg.as_tensor(X_NN), W1)), W2))
    m._rv("Bernoulli", "out", np.shape(Y_NN), sg._dist("Bernoulli", p=pt.sigmoid(pt.dot(act, w3))), None, Y_NN)
    return m


N_SV2 = 44
X_SURV = np.sin(np.arange(N_SV2) * 0.8) + 0.3 * np.cos(np.arange(N_SV2) * 2.1)
T_SURV = 0.4 + 1.8 * np.abs(np.sin(np.arange(N_SV2) * 1.27)) + 0.6 * (np.arange(N_SV2) % 3)
EVENT_SURV = ((np.arange(N_SV2) * 5) % 7 != 0).astype("float64")


def _weibull_censored(value, log_scale, shape_k, event):
    """The right-censored Weibull log-likelihood a user writes for `pm.CustomDist`: event * log h(t) + log S(t), h = k / lam (t / lam)^(k - 1),
    S = exp(-(t / lam)^k), lam = exp(log_scale)."""
    z = (pt.log(value) - log_scale) * shape_k
    return event * (pt.log(shape_k) - pt.log(value) + z) - pt.exp(z)


def survival_with_a_custom_density():
    """`pm.CustomDist(name, *params, logp=fn, observed=t)` (distributions/custom.py): the density is the USER's function of graph variables --
    a right-censored Weibull regression -- under improper priors (`pm.Flat`, `pm.HalfFlat`: continuous.py:364-443) and a `pm.Potential`."""
    m = sg.StubModel()
    b0 = m.Flat("b0")
    b1 = m.Normal("b1", 0.0, 1.0)
    k = m.HalfFlat("k")
    m.CustomDist("t", b0 + b1 * sg.as_tensor(X_SURV), k, sg.as_tensor(EVENT_SURV), logp=_weibull_censored, observed=T_SURV)
    m.Potential("k_prior", -0.5 * pt.sqr(pt.log(k)))
    return m


_raw_c = 0.9 * np.sin(np.arange(36) * 0.77) + 0.4 * np.cos(np.arange(36) * 1.9) + 0.2
Y_CENS_N = np.clip(_raw_c, -0.5, 0.8)                       # a Normal measurement with a detection floor and a saturation ceiling
_raw_e = 0.15 + 1.6 * np.abs(np.sin(np.arange(30) * 0.53)) ** 2
Y_CENS_E = np.minimum(_raw_e, 1.2)                          # waiting times followed up to t = 1.2


def censored_measurements():
    """`pm.Censored` (distributions/censored.py; logprob/censoring.py:198-250 `clip_logprob`): an interval-censored Normal (the density between
    the bounds, the base's `logcdf` at the floor, its `logccdf` at the ceiling) and a right-censored Exponential."""
    m = sg.StubModel()
    mu = m.Normal("mu", 0.0, 2.0)
    s = m.HalfNormal("s", 1.0)
    m.Censored("yn", ("Normal", dict(mu=mu, sigma=s)), -0.5, 0.8, observed=Y_CENS_N)
    lam = m.HalfNormal("lam", 2.0)
    m.Censored("ye", ("Exponential", dict(lam=lam)), None, 1.2, observed=Y_CENS_E)
    return m


MODELS = {
    "censored_measurements": censored_measurements,
    "survival_with_a_custom_density": survival_with_a_custom_density,
    "bayesian_neural_network": bayesian_neural_network,
    "double_well_sde": double_well_sde,
    "over_dispersed_counts": over_dispersed_counts,
    "multivariate_outcomes_lkj": multivariate_outcomes_lkj,
    "three_outcomes_lkj": three_outcomes_lkj,
    "correlated_outcomes_with_a_correlation_parameter": correlated_outcomes_with_a_correlation_parameter,
    "replicated_curves_under_a_squared_exponential_kernel": replicated_curves_under_a_squared_exponential_kernel,
    "three_outcomes_with_a_banded_precision_matrix": three_outcomes_with_a_banded_precision_matrix,
    "heavy_tailed_correlated_outcomes": heavy_tailed_correlated_outcomes,
    "skewed_measurements_and_discrete_lifetimes": skewed_measurements_and_discrete_lifetimes,
    "hurdle_models_of_positive_amounts": hurdle_models_of_positive_amounts,
    "disease_counts_over_a_lattice_of_areas": disease_counts_over_a_lattice_of_areas,
    "varying_slopes_lkj": varying_slopes_lkj,
    "three_correlated_effects_lkj": three_correlated_effects_lkj,
    "truncated_likelihoods": truncated_likelihoods,
    "ordered_probit_three_levels": ordered_probit_three_levels,
    "ordered_probit_four_levels": ordered_probit_four_levels,
    "zero_inflated_binomial_and_negative_binomial": zero_inflated_binomial_and_negative_binomial,
    "softmax_regression": softmax_regression,
    "robust_regression_with_dot": robust_regression_with_dot,
    "zero_sum_group_effects": zero_sum_group_effects,
    "zero_sum_log_rates": zero_sum_log_rates,
    "stochastic_volatility": stochastic_volatility,
    "ar2_with_constant": ar2_with_constant,
    "ar1_latent": ar1_latent,
    "random_walk_rate_under_counts": random_walk_rate_under_counts,
}
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "more_graphs.npz")
GOLDEN = os.path.join(HERE, "golden", "more_graphs_golden.npz")
