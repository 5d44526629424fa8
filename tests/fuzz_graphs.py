"""Drawn model GRAPHS for the lowering (tests/test_drawn_likelihood_graphs.py): the counterpart of tests/test_gpu_fuzz.py one level up.  That file
draws model SPECS (the engine against the oracle's interpreter); this one draws the graphs the lowering has to compile -- random
expressions over the model's variables in the `pytensor.tensor` vocabulary of tests/stubgraph.py (element-wise ops, comparisons under
`switch`, gathers, broadcasts between a vector of groups, a [G, D] matrix variable and the rows, reductions over a short axis, `clip`,
`maximum`, `logsumexp` over a short axis, concatenations) as the parameters of likelihoods whose densities are the reference's own
`logp` bodies, and as potentials.  Ground truth is torch autograd of the graph itself (tests/graph_torch.py): neither the lowering nor the
spec IR nor the oracle takes part in it.  Committed: tests/golden/fuzz_graphs.npz (the graphs) and fuzz_graphs_golden.npz, written by
tests/golden/make_fuzz_golden.py."""
import os

import numpy as np

import stubgraph as sg

pt = sg.pt
N_CASES = 52         # (cases 40 ..: 1 300 rows -- `pt.dot(X, b)` over that many rows is a linear predictor, dense node 5, not a written-out product)


def fuzz_graph_model(case: int):
    rg = np.random.default_rng(77000 + case)
    pick = lambda *xs: xs[int(rg.integers(len(xs)))]      # noqa: E731
    N = int(pick(12, 60, 250))
    if case >= 40:
        N = 1300
    G = int(pick(3, 7))
    D = int(pick(2, 3, 5))
    gi = rg.integers(0, G, size=N)
    X = rg.normal(size=(N, D))
    x1 = rg.normal(size=N)
    m = sg.StubModel()
    mu = m.Normal("mu", 0.0, 2.0)
    s = pick(lambda: m.HalfNormal("s", 1.5), lambda: m.Exponential("s", 1.0), lambda: m.HalfCauchy("s", 1.0), lambda: m.Gamma("s", 2.0, 1.5))()
    r = pick(lambda: m.Uniform("r", -1.0, 1.0), lambda: m.Beta("r", 2.0, 3.0))()
    a = m.Normal("a", mu, s, shape=(G,))
    W = m.Normal("W", 0.0, 1.0, shape=(G, D))
    b = m.Normal("b", 0.0, 1.0, shape=(D,))

    # ---- row-shaped building blocks (N elements each)
    def leaf():
        k = pick("gather", "slopes", "rowsum", "data", "scalar", "bdot") if case < 40 else pick("gather", "bdot", "bdot", "data", "scalar", "slopes")
        if k == "gather":
            return a[gi]
        if k == "slopes":                                   # varying slopes: a row of W per observation, times the covariates, summed
            return (W[gi] * sg.as_tensor(X)).sum(axis=-1)
        if k == "rowsum":                                   # a broadcast of b over the rows, reduced over the short axis
            return (sg.as_tensor(X) * b[None, :]).sum(axis=-1)
        if k == "data":
            return r * sg.as_tensor(x1)
        if k == "scalar":
            return mu * 0.3 + sg.as_tensor(x1) * 0.1
        return pt.dot(sg.as_tensor(X), b)                   # (a short inner dimension: written out)

    def unary(e):
        k = pick("tanh", "softplus", "sigmoid", "exp", "sqr", "abs", "log1p", "clip", "switch", "id", "id")
        if k == "tanh":
            return pt.tanh(e * 0.7)
        if k == "softplus":
            return pt.softplus(e)
        if k == "sigmoid":
            return pt.sigmoid(e) * 2.0 - 1.0
        if k == "exp":
            return pt.exp(e * 0.25)
        if k == "sqr":
            return pt.sqr(e) * 0.3
        if k == "abs":
            return pt.abs(e - 0.1)
        if k == "log1p":
            return pt.log1p(pt.sqr(e))
        if k == "clip":
            return pt.clip(e, -1.5, 1.2)
        if k == "switch":
            return pt.switch(pt.gt(e, 0.2), e * 0.5, pt.expm1(e * 0.3))
        return e

    def expr(depth):
        if depth == 0:
            return leaf()
        k = pick("un", "un", "add", "mul", "max", "lse")
        if k == "un":
            return unary(expr(depth - 1))
        x_, y_ = expr(depth - 1), expr(depth - 1)
        if k == "add":
            return x_ + y_ * 0.5
        if k == "mul":
            return x_ * pt.tanh(y_)
        if k == "max":
            return pt.maximum(x_, y_ - 0.3)
        return pt.logsumexp(pt.stack([x_, y_, x_ * 0.5 - y_], axis=-1), axis=-1)       # (three alternatives per row)

    loc = expr(int(pick(1, 2, 2, 3)))
    scale = pick(lambda: s, lambda: pt.exp(a[gi] * 0.2) * 0.7 + 0.1, lambda: pt.softplus(expr(1)) + 0.2)()
    lik = pick("Normal", "StudentT", "Cauchy", "Logistic", "Gumbel", "Poisson", "NegativeBinomial", "Bernoulli", "Gamma", "Weibull", "SkewNormal", "Laplace")
    eta = 0.4 * np.sin(gi) + 0.3 * x1
    if lik == "Normal":
        m.Normal("y", mu=loc, sigma=scale, observed=eta + 0.5 * rg.normal(size=N))
    elif lik == "StudentT":
        nu = m.Gamma("nu", 2.0, 0.1)
        m.StudentT("y", nu, mu=loc, sigma=scale, observed=eta + 0.5 * rg.standard_t(4, size=N))
    elif lik == "Cauchy":
        m.Cauchy("y", alpha=loc, beta=scale, observed=eta + 0.3 * rg.standard_cauchy(size=N))
    elif lik == "Logistic":
        m.Logistic("y", mu=loc, s=scale, observed=eta + 0.4 * rg.logistic(size=N))
    elif lik == "Gumbel":
        m.Gumbel("y", mu=loc, beta=scale, observed=eta + 0.4 * rg.gumbel(size=N))
    elif lik == "Poisson":
        m.Poisson("y", mu=pt.exp(loc * 0.3 + 0.4), observed=rg.poisson(np.exp(0.4 + 0.3 * eta)).astype("float64"))
    elif lik == "NegativeBinomial":
        alpha = m.Exponential("alpha", 0.5)
        m.NegativeBinomial("y", mu=pt.exp(loc * 0.3 + 0.4), alpha=alpha, observed=rg.poisson(np.exp(0.4 + 0.3 * eta) * rg.gamma(3.0, 1 / 3.0, size=N)).astype("float64"))
    elif lik == "Bernoulli":
        m.Bernoulli("y", logit_p=loc, observed=(rg.random(N) < 1.0 / (1.0 + np.exp(-eta))).astype("float64"))
    elif lik == "Gamma":
        k_ = m.Gamma("k", 2.0, 1.0)
        m.Gamma("y", alpha=k_, beta=k_ * pt.exp(-(loc * 0.3)), observed=rg.gamma(2.0, 0.6, size=N) + 0.05)
    elif lik == "Weibull":
        k_ = m.Gamma("k", 2.0, 1.0)
        m.Weibull("y", alpha=k_, beta=pt.exp(loc * 0.3), observed=rg.weibull(1.5, size=N) + 0.05)
    elif lik == "SkewNormal":
        sk = m.Normal("skew", 0.0, 2.0)
        m.SkewNormal("y", alpha=sk, mu=loc, sigma=scale, observed=eta + 0.5 * rg.normal(size=N))
    else:
        m.Laplace("y", mu=loc, b=scale, observed=eta + 0.4 * rg.laplace(size=N))
    if rg.random() < 0.4:
        m.Potential("pen", -0.05 * pt.sum(pt.sqr(W), axis=-1) * pt.sqr(r))           # (a [G]-shaped potential: a reduction over the short axis)
    if rg.random() < 0.3:
        m.Potential("soft_order", pt.log(pt.sigmoid((a[1:] - a[:-1]) * 2.0)))         # (slices of a vector variable)
    return m


MODELS = {f"case_{c:02d}": (lambda c=c: fuzz_graph_model(c)) for c in range(N_CASES)}
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "fuzz_graphs.npz")
GOLDEN = os.path.join(HERE, "golden", "fuzz_graphs_golden.npz")
