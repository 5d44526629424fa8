"""The op-by-op models of tests/lowering_models.py (`GENERAL`: no `ModelBuilder` twin) written down a SECOND time, with SciPy's
distributions and NumPy, straight from the model descriptions -- nothing of the graph stand-in, the reference's `logp` bodies, the
lowering or the oracle is involved.  Compared with the committed goldens (torch autograd of the reference-built graphs,
tests/golden/general_graphs_golden.npz) at the goldens' own points.

Why: the `logp` bodies in those graphs are the reference's source, but `Model.logp`'s assembly around them is the stand-in's own
(tests/stubgraph.py), and goldens made from the stand-in cannot catch the stand-in -- DESIGN.md section 2 tells of the one time that
mattered (`transforms.ordered`, found by a restatement like these).  Jacobians of the value transforms are taken numerically
(`slogdet` of the backward map's Jacobian by central differences), so not even their formulas are shared."""
import os
import sys

import numpy as np
import pytest
from scipy import special, stats

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lowering_models as lm  # noqa: E402


# ---- value transforms: backward maps written from their definitions; log|det J| by finite differences -------------------------------
def _log(v):
    return np.exp(v)


def _logodds(v):
    return special.expit(v)


def _interval(lo, hi):
    return lambda v: lo + (hi - lo) * special.expit(v)


def _simplex(v):           # K - 1 free values -> a point of the simplex (softmax of [v, -sum v])
    e = np.concatenate([v, [-v.sum()]])
    w = np.exp(e - e.max())
    return w / w.sum()


def _ordered(v):
    return np.cumsum(np.concatenate([v[:1], np.exp(v[1:])]))


def _logjac(back, v, free=None):
    """log |det d back(v)[free] / d v| -- `free`: which constrained coordinates count (a simplex point has one coordinate too many)."""
    v = np.atleast_1d(np.asarray(v, dtype="float64"))
    n = v.size
    J = np.empty((n, n))
    for j in range(n):
        h = 1e-6 * max(1.0, abs(v[j]))
        e = np.zeros(n)
        e[j] = h
        hi, lo = np.atleast_1d(back(v + e)), np.atleast_1d(back(v - e))
        J[:, j] = ((hi - lo) / (2 * h))[:n] if free is None else ((hi - lo) / (2 * h))[free]
    return np.linalg.slogdet(J)[1]


class _Point:
    """Reads a raveled point variable by variable: `x = p.take(size, backward)` returns the constrained value and adds the Jacobian."""

    def __init__(self, q):
        self.q, self.at, self.jac = np.asarray(q, dtype="float64"), 0, 0.0

    def take(self, size=1, back=None, elementwise=True, free=None):
        v = self.q[self.at : self.at + size]
        self.at += size
        if back is None:
            return v if size > 1 else v[0]
        if elementwise:
            self.jac += sum(_logjac(back, [vi]) for vi in v)
            x = back(v)
        else:
            self.jac += _logjac(back, v, free)
            x = back(v)
        return x if np.size(x) > 1 else float(np.ravel(x)[0])

    def done(self):
        assert self.at == self.q.size
        return self.jac


def _halfnormal(x, s):
    return stats.halfnorm(scale=s).logpdf(x)


def _zi(y, psi, base):
    return np.where(y == 0, np.log((1 - psi) + psi * base.pmf(0)), np.log(psi) + base.logpmf(y))


# ---- the models ------------------------------------------------------------------------------------------------------------------
def robust_regression(q):
    p = _Point(q)
    a, b, sigma, nu = p.take(), p.take(), p.take(1, _log), p.take(1, _log)
    lp = stats.norm(0, 2).logpdf([a, b]).sum() + _halfnormal(sigma, 1.0) + stats.gamma(2.0, scale=1 / 0.1).logpdf(nu)
    return lp + stats.t(nu, a + b * lm.XGEN, sigma).logpdf(lm.YGEN).sum() + p.done()


def random_shape_parameters(q):
    p = _Point(q)
    al, be, a_, b_, pp = p.take(1, _log), p.take(1, _log), p.take(1, _log), p.take(1, _log), p.take(1, _logodds)
    lp = _halfnormal(al, 2.0) + stats.expon(scale=1.0).logpdf(be) + stats.gamma(2.0, scale=1.0).logpdf(a_) + stats.gamma(3.0, scale=1 / 1.5).logpdf(b_)
    lp += stats.beta(a_, b_).logpdf(pp)
    lp += stats.gamma(al, scale=1 / be).logpdf(lm.WPOS).sum() + stats.invgamma(al, scale=0.7).logpdf(lm.WPOS).sum()
    lp += stats.binom(lm.NN.astype(int), pp).logpmf(lm.CNT).sum() + stats.beta(a_, 2.5).logpdf(lm.UNIT).sum()
    return lp + p.done()


def negative_binomial_regression(q):
    p = _Point(q)
    a, b, alpha = p.take(), p.take(), p.take(1, _log)
    mu = np.exp(a + b * lm.XGEN[:25])
    lp = stats.norm(1, 1).logpdf(a) + stats.norm(0, 1).logpdf(b) + stats.expon(scale=1 / 0.5).logpdf(alpha)
    return lp + stats.nbinom(alpha, alpha / (alpha + mu)).logpmf(lm.CNT2).sum() + p.done()


def density_zoo(q):
    p = _Point(q)
    k, lam, loc, s, sk, a_, pg = p.take(1, _log), p.take(1, _log), p.take(), p.take(1, _log), p.take(), p.take(1, _log), p.take(1, _logodds)
    lp = _halfnormal(k, 2.0) + _halfnormal(lam, 2.0) + stats.norm(0, 2).logpdf(loc) + _halfnormal(s, 1.0) + stats.norm(0, 2).logpdf(sk)
    lp += _halfnormal(a_, 2.0) + stats.beta(2, 2).logpdf(pg)
    lp += stats.weibull_min(k, scale=lam).logpdf(lm.WPOS).sum() + stats.logistic(loc, s).logpdf(lm.YGEN[:10]).sum()
    lp += stats.gumbel_r(loc, s).logpdf(lm.YGEN[10:20]).sum() + stats.skewnorm(sk, loc, s).logpdf(lm.YGEN[20:]).sum()
    lp += stats.betabinom(lm.NN.astype(int), a_, 2.0).logpmf(lm.CNT).sum() + stats.geom(pg).logpmf(lm.CNT + 1.0).sum()
    return lp + stats.norm(lam ** k, 1.0).logpdf(np.full(3, 0.8)).sum() + p.done()


def density_zoo_2(q):
    p = _Point(q)
    mu, lam, a, b, kap, loc, sc = p.take(1, _log), p.take(1, _log), p.take(1, _log), p.take(1, _log), p.take(1, _log), p.take(), p.take(1, _log)
    lp = _halfnormal(mu, 2.0) + _halfnormal(lam, 3.0) + _halfnormal(a, 2.0) + _halfnormal(b, 2.0) + _halfnormal(kap, 1.5) + stats.norm(0, 2).logpdf(loc)
    lp += _halfnormal(sc, 1.5)
    lp += stats.invgauss(mu / lam, scale=lam).logpdf(lm._YPOS).sum()
    x = lm._YUNIT
    lp += (np.log(a) + np.log(b) + (a - 1) * np.log(x) + (b - 1) * np.log1p(-x ** a)).sum()                 # Kumaraswamy
    lp += stats.laplace_asymmetric(kap, loc=loc, scale=1 / sc).logpdf(lm.YGEN[:12]).sum()
    return lp + stats.moyal(loc, sc).logpdf(lm.YGEN[5:17]).sum() + p.done()


def density_zoo_3(q):
    p = _Point(q)
    loc, sc, al, nu, en, cc = p.take(), p.take(1, _log), p.take(1, _log), p.take(1, _log), p.take(1, _log), p.take(1, _logodds)
    lp = stats.norm(0, 2).logpdf(loc) + _halfnormal(sc, 1.5) + _halfnormal(al, 3.0) + stats.gamma(2.0, scale=1 / 0.3).logpdf(nu) + _halfnormal(en, 2.0)
    lp += stats.beta(2, 2).logpdf(cc)
    lp += stats.pareto(al, scale=0.25).logpdf(lm._YPOS).sum()
    lp += (np.log(2.0) + stats.t(nu, 0.0, sc).logpdf(lm._YPOS)).sum()                                      # HalfStudentT
    lp += stats.exponnorm(en / sc, loc=loc, scale=sc).logpdf(lm.YGEN[12:]).sum()                            # ExGaussian(mu, sigma, nu)
    return lp + stats.triang(cc, loc=0.0, scale=1.0).logpdf(lm._YUNIT).sum() + p.done()


def ordinal_regression(q):
    p = _Point(q)
    a, c0, d1, d2 = p.take(), p.take(), p.take(), p.take()
    lp = stats.norm(0, 2).logpdf(a) + stats.norm(-1, 2).logpdf(c0) + stats.norm(0, 1).logpdf([d1, d2]).sum()
    cut = np.array([c0, c0 + np.exp(d1), c0 + np.exp(d1) + np.exp(d2)])
    cum = special.expit(cut[None, :] - (a * lm._XORD)[:, None])                                            # P(y <= k)
    pr = np.diff(np.concatenate([np.zeros((40, 1)), cum, np.ones((40, 1))], axis=1), axis=1)
    return lp + np.log(pr[np.arange(40), lm._YORD.astype(int)]).sum() + p.done()


def zero_inflated_poisson(q):
    p = _Point(q)
    psi, a, b = p.take(1, _logodds), p.take(), p.take()
    lp = stats.beta(2, 2).logpdf(psi) + stats.norm(1, 1).logpdf(a) + stats.norm(0, 1).logpdf(b)
    mu = np.exp(a + b * np.linspace(-1.0, 1.0, 36))
    return lp + _zi(lm._YZIP, psi, stats.poisson(mu)).sum() + p.done()


def mixture_with_ordered_means(q):
    p = _Point(q)
    mu, sigma = p.take(3, _ordered, elementwise=False), p.take(3, _log)
    w = p.take(2, _simplex, elementwise=False, free=[0, 1])
    lp = stats.norm([-1.0, 0.0, 1.0], 5.0).logpdf(mu).sum() + _halfnormal(sigma, 2.0).sum() + stats.dirichlet(np.ones(3)).logpdf(w)
    y = lm._YMIXO
    return lp + special.logsumexp(np.log(w)[None, :] + stats.norm(mu[None, :], sigma[None, :]).logpdf(y[:, None]), axis=1).sum() + p.done()


def hierarchical_regression_noncentred(q):
    p = _Point(q)
    mu, sigma, z, s, alpha = p.take(), p.take(1, _log), p.take(7), p.take(1, _log), p.take()
    lp = stats.norm(0, 1).logpdf(mu) + _halfnormal(sigma, 1.0) + stats.norm(0, 1).logpdf(z).sum() + _halfnormal(s, 1.0) + stats.norm(0, 2).logpdf(alpha)
    return lp + stats.norm(alpha + lm.XH @ (mu + sigma * z), s).logpdf(lm.YH).sum() + p.done()


def hierarchical_logistic_vector_hyper(q):
    p = _Point(q)
    mu, sigma, z = p.take(7), p.take(7, _log), p.take(7)
    lp = stats.norm(0, 1).logpdf(mu).sum() + _halfnormal(sigma, 1.0).sum() + stats.norm(0, 1).logpdf(z).sum()
    return lp + stats.bernoulli(special.expit(lm.XH @ (mu + sigma * z))).logpmf(lm.YHB).sum() + p.done()


def glm_with_mvnormal_prior(q):
    beta = np.asarray(q)
    return stats.multivariate_normal(np.zeros(7), lm.COV7).logpdf(beta) + stats.bernoulli(special.expit(lm.XH @ beta)).logpmf(lm.YHB).sum()


def shapes_broadcast_gather_and_rowsum(q):
    p = _Point(q)
    z2, s3 = p.take(6).reshape(2, 3), p.take(3, _log)
    m0, sd, zz = p.take(), p.take(1, _log), p.take(4)
    mu, sigma, z = p.take(3), p.take(3, _log), p.take(15).reshape(5, 3)
    lp = stats.norm(0, 1).logpdf(z2).sum() + _halfnormal(s3, 1.0).sum() + stats.norm(z2 * s3, 1.0).logpdf(np.arange(6.0).reshape(2, 3) * 0.1).sum()
    lp += stats.norm(0, 1).logpdf(m0) + _halfnormal(sd, 1.0) + stats.norm(0, 1).logpdf(zz).sum()
    lp += stats.norm((m0 + sd * zz)[lm._GI6] * lm._XS10, 0.7).logpdf(lm._YS10).sum()
    lp += stats.norm(0, 1).logpdf(mu).sum() + _halfnormal(sigma, 1.0).sum() + stats.norm(0, 1).logpdf(z).sum()
    eta = (lm._XP * (mu + sigma * z)[lm._GP]).sum(axis=1)
    return lp + stats.poisson(np.exp(0.3 * eta)).logpmf(lm._YP).sum() + p.done()


def dirichlet_multinomial(q):
    p = _Point(q)
    w = p.take(3, _simplex, elementwise=False, free=[0, 1, 2])
    w2 = p.take(1, _simplex, elementwise=False, free=[0])
    lp = stats.dirichlet([1.5, 2.0, 3.0, 0.7]).logpdf(w) + stats.multinomial(20, w).logpmf([5, 7, 6, 2])
    return lp + stats.dirichlet([1.5, 2.0]).logpdf(w2) + stats.multinomial(9, w2).logpmf([5, 4]) + p.done()


def mixtures_of_other_families(q):
    p = _Point(q)
    w = p.take(1, _simplex, elementwise=False, free=[0])
    lam1, lam2 = p.take(1, _log), p.take(1, _log)
    w2 = p.take(1, _simplex, elementwise=False, free=[0])
    mu, a, b = p.take(), p.take(3, _log), p.take(3, _log)
    lp = stats.dirichlet([1.0, 1.0]).logpdf(w) + stats.expon(scale=1.0).logpdf(lam1) + stats.expon(scale=1 / 0.2).logpdf(lam2)
    lp += stats.dirichlet([2.0, 1.0]).logpdf(w2) + stats.norm(0, 1).logpdf(mu) + _halfnormal(a, 3.0).sum() + _halfnormal(b, 2.0).sum()
    lse = special.logsumexp
    y = lm._YMIXP
    lp += lse(np.log(w)[None, :] + stats.poisson(np.array([lam1, lam2])[None, :]).logpmf(y[:, None]), axis=1).sum()
    y = lm._YMIXT
    lp += lse(np.stack([np.log(w2[0]) + stats.norm(mu, 1.0).logpdf(y), np.log(w2[1]) + stats.t(4.0, mu, 2.5).logpdf(y)], axis=1), axis=1).sum()
    y = lm._YMIXG
    lp += lse(np.log([0.45, 0.35, 0.2])[None, :] + stats.gamma(a[None, :], scale=1 / b[None, :]).logpdf(y[:, None]), axis=1).sum()
    return lp + p.done()


def categorical_free_standing(q):
    p = _Point(q)
    w = p.take(2, _simplex, elementwise=False, free=[0, 1])
    logits = p.take(4)
    cobs = np.array([0, 1, 2, 1, 1, 0, 2, 1, 1, 0, 1, 2])
    c = np.array([3, 0, 1, 1, 2, 3, 3, 0, 1])                                   # the discrete variable stands at its initial value
    lp = stats.dirichlet([1.0, 2.0, 1.5]).logpdf(w) + np.log(w[cobs]).sum() + stats.norm(0, 1.5).logpdf(logits).sum()
    return lp + (logits - special.logsumexp(logits))[c].sum() + p.done()


RESTATED = {f.__name__: f for f in (robust_regression, random_shape_parameters, negative_binomial_regression, density_zoo, density_zoo_2, density_zoo_3,
                                    ordinal_regression, zero_inflated_poisson, mixture_with_ordered_means, hierarchical_regression_noncentred,
                                    hierarchical_logistic_vector_hyper, glm_with_mvnormal_prior, shapes_broadcast_gather_and_rowsum,
                                    dirichlet_multinomial, mixtures_of_other_families, categorical_free_standing)}


def test_every_twinless_model_is_restated():
    assert sorted(RESTATED) == sorted(lm.GENERAL)


@pytest.mark.parametrize("name", sorted(RESTATED))
def test_the_committed_goldens_are_the_textbook_densities(name):
    z = np.load(lm.GENERAL_GOLDEN)
    for q, lp0 in zip(z[f"{name}__q"], z[f"{name}__logp"]):
        want = RESTATED[name](q)
        # (the Jacobians are finite differences: 1e-7 of the value is their accuracy, far below any mistake worth the name)
        assert abs(lp0 - want) <= 2e-7 * max(1.0, abs(want)), (name, lp0, want)
