"""A stand-in for the slice of PyTensor / PyMC that `pymc_amd.lowering.lower_to_spec` walks (PyTensor cannot be imported in
the build image).  TEST INFRASTRUCTURE.

* graph protocol: `Variable(owner, name)`, `Apply(op, inputs)`, ops named as PyTensor names them (`Elemwise` with a
  `scalar_op` object whose class is `Add`, `Mul`, `Sub`, `TrueDiv`, `Pow`, `Exp`, `Log`, `Log1p`, `Sqrt`, `Neg`, `Switch`, `GE`,
  `GT`, `LT`, `LE`, `EQ`, `NEQ`, `OR`, `AND`, `Sigmoid`, `Softplus`, `GammaLn`, `Reciprocal`, `Sign`, `Erf`, `Erfc`, `Erfcx`, `Sqr`, `Second`; `DimShuffle`; `Sum` with `.axis`; `AdvancedSubtensor1`; `CheckParameterValue`; constants carry
  `.data`), with operator overloading so that the distribution code below reads like the reference's;
* distributions: `logp` bodies TRANSCRIBED from the reference, each citing its lines -- these build exactly the expression a
  `pm.Model` would hand to the compiler BEFORE rewrites (`Model.logp`, model/core.py:612-695);
* `StubModel`: `value_vars`, value transforms (`rvs_to_transforms`), `logp(sum=False)` in the reference's order (free RVs,
  observed RVs, potentials) with the Jacobian term added to transformed variables' own factors
  (`transformed_conditional_logp`, logprob/basic.py:618-667).
"""
import numpy as np


class Apply:
    def __init__(self, op, inputs):
        self.op, self.inputs = op, list(inputs)


class _Type:
    def __init__(self, shape):
        self.shape = tuple(shape)


class Variable:
    def __init__(self, owner=None, name=None, shape=()):
        self.owner, self.name, self.type = owner, name, _Type(shape)

    def _bin(self, other, cls, swap=False):
        other = as_tensor(other)
        a, b = (other, self) if swap else (self, other)
        return elemwise(cls, a, b)

    def __add__(self, o): return self._bin(o, Add)
    def __radd__(self, o): return self._bin(o, Add, True)
    def __sub__(self, o): return self._bin(o, Sub)
    def __rsub__(self, o): return self._bin(o, Sub, True)
    def __mul__(self, o): return self._bin(o, Mul)
    def __rmul__(self, o): return self._bin(o, Mul, True)
    def __truediv__(self, o): return self._bin(o, TrueDiv)
    def __rtruediv__(self, o): return self._bin(o, TrueDiv, True)
    def __neg__(self): return elemwise(Neg, self)
    def __gt__(self, o): return self._bin(o, GT)
    def __ge__(self, o): return self._bin(o, GE)
    def __lt__(self, o): return self._bin(o, LT)
    def __le__(self, o): return self._bin(o, LE)
    def __pow__(self, o): return self._bin(o, Pow)
    def __getitem__(self, idx): return Variable(Apply(AdvancedSubtensor1(), [self, as_tensor(idx)]), shape=(len(np.asarray(idx)),) + self.type.shape[1:])
    def sum(self, axis=None): return Variable(Apply(Sum(axis), [self]), shape=())


class TensorConstant(Variable):
    def __init__(self, data):
        data = np.asarray(data)
        super().__init__(None, None, data.shape)
        self.data = data


def as_tensor(x):
    return x if isinstance(x, Variable) else TensorConstant(x)


# ---- ops (class names are what the walker keys on) ----
class Elemwise:
    def __init__(self, scalar_op):
        self.scalar_op = scalar_op


class DimShuffle:
    pass


class AdvancedSubtensor1:
    pass


class Sum:
    def __init__(self, axis):
        self.axis = axis


class CheckParameterValue:
    def __init__(self, msg):
        self.msg = msg


class All:
    pass


for _n in ("Add", "Sub", "Mul", "TrueDiv", "Pow", "Exp", "Log", "Log1p", "Sqrt", "Neg", "Switch", "GE", "GT", "LT", "LE", "EQ", "OR", "AND", "Sigmoid", "Abs",
           "GammaLn", "Reciprocal", "Sign", "NEQ", "Second", "Softplus", "Erf", "Erfc", "Erfcx", "Sqr"):
    globals()[_n] = type(_n, (), {})


class _ScalarVar:
    """A scalar variable of a `Composite`'s inner graph (`pytensor.scalar`): `.owner.op` is the scalar op itself."""

    def __init__(self, owner=None, data=None):
        self.owner = owner
        if data is not None:
            self.data = np.asarray(data)


class Composite:
    """`pytensor.scalar.basic.Composite`: a fused scalar sub-graph, reduced to `inputs`, `outputs` and the inner nodes' protocol.
    `build(n_in, fn)`: `fn` receives scalar placeholders and composes them with `Composite.op(cls, *args)`."""

    def __init__(self, inputs, outputs):
        self.inputs, self.outputs = list(inputs), list(outputs)
        self.fgraph = self

    @staticmethod
    def op(cls, *args):
        args = [a if isinstance(a, _ScalarVar) else _ScalarVar(data=a) for a in args]
        return _ScalarVar(owner=Apply(cls(), args))

    @classmethod
    def build(cls, n_in, fn):
        ins = [_ScalarVar() for _ in range(n_in)]
        return cls(ins, [fn(*ins)])


def fused(comp, *ins):
    """`Elemwise(Composite)(*ins)`."""
    ins = [as_tensor(i) for i in ins]
    return Variable(Apply(Elemwise(comp), ins), shape=_bshape(*ins))


def _bshape(*vs):
    return np.broadcast_shapes(*[v.type.shape for v in vs])


def _dimshuffle_to(v, shape):
    """PyTensor inserts a DimShuffle wherever an operand needs broadcast dimensions."""
    if v.type.shape == tuple(shape) or isinstance(v, TensorConstant) and v.data.ndim == 0:
        return v
    return Variable(Apply(DimShuffle(), [v]), shape=shape)


def elemwise(cls, *ins):
    ins = [as_tensor(i) for i in ins]
    shape = _bshape(*ins)
    ins = [_dimshuffle_to(i, shape) if i.type.shape != () else i for i in ins]
    return Variable(Apply(Elemwise(cls()), ins), shape=shape)


class pt:   # the `pytensor.tensor` names the reference's logp bodies use
    pow = staticmethod(lambda a, b: elemwise(Pow, a, b))
    log = staticmethod(lambda a: elemwise(Log, a))
    log1p = staticmethod(lambda a: elemwise(Log1p, a))
    sqrt = staticmethod(lambda a: elemwise(Sqrt, a))
    exp = staticmethod(lambda a: elemwise(Exp, a))
    sigmoid = staticmethod(lambda a: elemwise(Sigmoid, a))
    switch = staticmethod(lambda c, a, b: elemwise(Switch, c, a, b))
    ge = staticmethod(lambda a, b: elemwise(GE, a, b))
    lt = staticmethod(lambda a, b: elemwise(LT, a, b))
    gt = staticmethod(lambda a, b: elemwise(GT, a, b))
    or_ = staticmethod(lambda a, b: elemwise(OR, a, b))
    abs = staticmethod(lambda a: elemwise(Abs, a))
    eq = staticmethod(lambda a, b: elemwise(EQ, a, b))
    le = staticmethod(lambda a, b: elemwise(LE, a, b))
    and_ = staticmethod(lambda a, b: elemwise(AND, a, b))
    bitwise_and = staticmethod(lambda a, b: elemwise(AND, a, b))
    gammaln = staticmethod(lambda a: elemwise(GammaLn, a))
    reciprocal = staticmethod(lambda a: elemwise(Reciprocal, a))
    sign = staticmethod(lambda a: elemwise(Sign, a))
    neq = staticmethod(lambda a, b: elemwise(NEQ, a, b))
    where = staticmethod(lambda c, a, b: elemwise(Switch, c, a, b))
    fill = staticmethod(lambda a, b: elemwise(Second, a, b))
    softplus = staticmethod(lambda a: elemwise(Softplus, a))
    erf = staticmethod(lambda a: elemwise(Erf, a))
    erfc = staticmethod(lambda a: elemwise(Erfc, a))
    erfcx = staticmethod(lambda a: elemwise(Erfcx, a))
    sqr = staticmethod(lambda a: elemwise(Sqr, a))
    square = staticmethod(lambda a: elemwise(Sqr, a))
    inf = np.inf


gammaln = pt.gammaln


def check_parameters(expr, *conds, msg=""):   # distributions/dist_math.py:50-74
    allc = Variable(Apply(All(), [as_tensor(c) for c in conds]), shape=())
    return Variable(Apply(CheckParameterValue(msg), [expr, allc]), shape=expr.type.shape)


# ---- logp bodies, transcribed ----
def normal_logp(value, mu, sigma):          # distributions/continuous.py:526-532
    res = -0.5 * pt.pow((value - mu) / sigma, 2) - pt.log(pt.sqrt(2.0 * np.pi)) - pt.log(sigma)
    return check_parameters(res, sigma > 0, msg="sigma > 0")


def halfnormal_logp(value, loc, sigma):     # distributions/continuous.py:909-916
    res = -0.5 * pt.pow((value - loc) / sigma, 2) + pt.log(pt.sqrt(2.0 / np.pi)) - pt.log(sigma)
    res = pt.switch(pt.ge(value, loc), res, -np.inf)
    return check_parameters(res, sigma > 0, msg="sigma > 0")


def cauchy_logp(value, alpha, beta):        # distributions/continuous.py:2287-2293
    res = -pt.log(np.pi) - pt.log(beta) - pt.log1p(pt.pow((value - alpha) / beta, 2))
    return check_parameters(res, beta > 0, msg="beta > 0")


def halfcauchy_logp(value, beta):           # distributions/continuous.py:2383-2390
    res = pt.log(2) + cauchy_logp(value, 0, beta)
    res = pt.switch(value >= 0, res, -np.inf)
    return check_parameters(res, beta > 0, msg="beta > 0")


def laplace_logp(value, mu, b):             # distributions/continuous.py:1570-1576
    res = -pt.log(2 * b) - pt.abs(value - mu) / b
    return check_parameters(res, b > 0, msg="b > 0")


def lognormal_logp(value, mu, sigma):       # distributions/continuous.py:1807-1821
    res = -0.5 * pt.pow((pt.log(value) - mu) / sigma, 2) - 0.5 * pt.log(2.0 * np.pi) - pt.log(sigma) - pt.log(value)
    res = pt.switch(pt.gt(value, 0.0), res, -np.inf)
    return check_parameters(res, sigma > 0, msg="sigma > 0")


def logpow(x, m):                           # distributions/dist_math.py:92-107
    log_x = pt.log(x)
    return pt.switch(pt.and_(pt.eq(log_x, -np.inf), pt.le(m, 0)), pt.switch(pt.eq(m, 0), 0.0, -np.inf), m * log_x)


def factln(n):                              # distributions/dist_math.py:110-111
    return gammaln(n + 1)


def get_tau_sigma(sigma):                   # distributions/continuous.py:234-239 (the `tau is None` branch)
    sigma = as_tensor(sigma)
    return (sigma ** -2.0) * pt.sign(sigma), sigma


def studentt_logp(value, nu, mu, sigma):    # distributions/continuous.py:1935-1950
    lam, _ = get_tau_sigma(sigma=sigma)
    res = (gammaln((nu + 1.0) / 2.0) + 0.5 * pt.log(lam / (nu * np.pi)) - gammaln(nu / 2.0)
           - (nu + 1.0) / 2.0 * pt.log1p(lam * (value - mu) ** 2 / nu))
    return check_parameters(res, lam > 0, nu > 0, msg="lam > 0, nu > 0")


def beta_logp(value, alpha, beta):          # distributions/continuous.py:1248-1262
    res = (pt.switch(pt.eq(alpha, 1.0), 0.0, (alpha - 1.0) * pt.log(value))
           + pt.switch(pt.eq(beta, 1.0), 0.0, (beta - 1.0) * pt.log1p(-value))
           - (pt.gammaln(alpha) + pt.gammaln(beta) - pt.gammaln(alpha + beta)))
    res = pt.switch(pt.bitwise_and(pt.ge(value, 0.0), pt.le(value, 1.0)), res, -np.inf)
    return check_parameters(res, alpha > 0, beta > 0, msg="alpha > 0, beta > 0")


def gamma_logp(value, alpha, scale):        # distributions/continuous.py:2512-2521 (`scale = reciprocal(beta)`, :2484-2492)
    beta = pt.reciprocal(scale)
    res = -pt.gammaln(alpha) + logpow(beta, alpha) - beta * value + logpow(value, alpha - 1)
    res = pt.switch(pt.ge(value, 0.0), res, -np.inf)
    return check_parameters(res, alpha > 0, beta > 0, msg="alpha > 0, beta > 0")


def invgamma_logp(value, alpha, beta):      # distributions/continuous.py:2631-2639
    res = -pt.gammaln(alpha) + logpow(beta, alpha) - beta / value + logpow(value, -alpha - 1)
    res = pt.switch(pt.ge(value, 0.0), res, -np.inf)
    return check_parameters(res, alpha > 0, beta > 0, msg="alpha > 0, beta > 0")


def poisson_logp(value, mu):                # distributions/discrete.py:581-597
    res = pt.switch(pt.lt(value, 0), -np.inf, logpow(mu, value) - factln(value) - mu)
    res = pt.switch(pt.eq(mu, 0) * pt.eq(value, 0), 0, res)
    return check_parameters(res, mu >= 0, msg="mu >= 0")


def binomln(n, k):                          # distributions/dist_math.py:114-115
    return factln(n) - factln(k) - factln(n - k)


def uniform_logp(value, lower, upper):      # distributions/continuous.py:309-321
    res = pt.switch(pt.bitwise_and(pt.ge(value, lower), pt.le(value, upper)), pt.fill(value, -pt.log(upper - lower)), -np.inf)
    return check_parameters(res, lower <= upper, msg="lower <= upper")


def binomial_logp(value, n, p):             # distributions/discrete.py:141-154
    res = pt.switch(pt.or_(pt.lt(value, 0), pt.gt(value, n)), -np.inf, binomln(n, value) + logpow(p, value) + logpow(1 - p, n - value))
    return check_parameters(res, n >= 0, 0 <= p, p <= 1, msg="n >= 0, 0 <= p <= 1")


def interval_backward(value, a, b):         # logprob/transforms.py:1026-1053 (both bounds given)
    a, b = as_tensor(a), as_tensor(b)
    exp_value = pt.exp(value)
    sigmoid_x = pt.sigmoid(value)
    lower_distance = exp_value + a
    upper_distance = b - exp_value
    return pt.where(pt.and_(pt.neq(a, -pt.inf), pt.neq(b, pt.inf)), sigmoid_x * b + (1 - sigmoid_x) * a,
                    pt.where(pt.neq(a, -pt.inf), lower_distance, pt.where(pt.neq(b, pt.inf), upper_distance, value)))


def interval_log_jac_det(value, a, b):      # logprob/transforms.py:1055-1070
    a, b = as_tensor(a), as_tensor(b)
    s = pt.softplus(-value)
    return pt.where(pt.and_(pt.neq(a, -pt.inf), pt.neq(b, pt.inf)), pt.log(b - a) - 2 * s - value,
                    pt.where(pt.or_(pt.neq(a, -pt.inf), pt.neq(b, pt.inf)), value, value * 0.0))


def normal_lcdf(mu, sigma, x):              # distributions/dist_math.py:126-133
    z = (x - mu) / sigma
    return pt.switch(pt.lt(z, -1.0), pt.log(pt.erfcx(-z / pt.sqrt(2.0)) / 2.0) - pt.sqr(z) / 2.0, pt.log1p(-pt.erfc(z / pt.sqrt(2.0)) / 2.0))


def normal_lccdf(mu, sigma, x):             # distributions/dist_math.py:136-142
    z = (x - mu) / sigma
    return pt.switch(pt.gt(z, 1.0), pt.log(pt.erfcx(z / pt.sqrt(2.0)) / 2.0) - pt.sqr(z) / 2.0, pt.log1p(-pt.erfc(-z / pt.sqrt(2.0)) / 2.0))


def log_diff_normal_cdf(mu, sigma, x, y):   # distributions/dist_math.py:145-183
    x = (x - mu) / sigma / pt.sqrt(2.0)
    y = (y - mu) / sigma / pt.sqrt(2.0)
    return pt.log(0.5) + pt.switch(
        pt.gt(y, 0),
        -pt.square(y) + pt.log(pt.erfcx(y) - pt.exp(pt.square(y) - pt.square(x)) * pt.erfcx(x)),
        pt.switch(pt.lt(x, 0), -pt.square(x) + pt.log(pt.erfcx(-x) - pt.exp(pt.square(x) - pt.square(y)) * pt.erfcx(-y)),
                  pt.log(pt.erf(x) - pt.erf(y))))


def truncnormal_logp(value, mu, sigma, lower, upper):   # distributions/continuous.py:720-746 (`None` = unbounded on that side)
    lb, ub = lower is not None, upper is not None
    if lb and ub:
        norm = log_diff_normal_cdf(mu, sigma, upper, lower)
    elif lb:
        norm = normal_lccdf(mu, sigma, lower)
    elif ub:
        norm = normal_lcdf(mu, sigma, upper)
    else:
        norm = 0.0
    logp = normal_logp(value, mu, sigma) - norm          # `_logprob_helper(Normal.dist(mu, sigma), value)`
    if lb:
        logp = pt.switch(value < lower, -np.inf, logp)
    if ub:
        logp = pt.switch(value > upper, -np.inf, logp)
    if lb and ub:
        logp = check_parameters(logp, pt.le(lower, upper), msg="lower_bound <= upper_bound")
    return logp


def bernoulli_logp(value, p):               # distributions/discrete.py:362-374
    res = pt.switch(pt.or_(pt.lt(value, 0), pt.gt(value, 1)), -np.inf, pt.switch(value, pt.log(p), pt.log1p(-p)))
    return check_parameters(res, 0 <= p, p <= 1, msg="0 <= p <= 1")


class _RV:
    def __init__(self, name, shape, logp_fn, params, transform=None, observed=None, bounds=None):
        self.name, self.shape, self.logp_fn, self.params, self.transform, self.observed = name, tuple(shape), logp_fn, params, transform, observed
        self.bounds = bounds
        if observed is None:
            vname = name if transform is None else f"{name}_{transform}__"   # util.py:138-155
            self.value = Variable(None, vname, self.shape)
            # what the rest of the graph sees in place of the RV: transform.backward(value) (logprob/transforms.py:880-891, 1026-1053, 1076-1088)
            self.expr = {None: self.value, "log": pt.exp(self.value) if transform == "log" else None,
                         "logodds": pt.sigmoid(self.value) if transform == "logodds" else None,
                         "interval": interval_backward(self.value, *bounds) if transform == "interval" else None}[transform]
        else:
            self.value, self.expr = None, TensorConstant(np.asarray(observed, dtype="float64"))


class StubModel:
    """`with pm.Model(): ...` reduced to what the lowering reads."""

    def __init__(self):
        self.free, self.obs, self.pots = [], [], []

    def _add(self, rv):
        (self.free if rv.observed is None else self.obs).append(rv)
        return rv.expr

    def Normal(self, name, mu=0.0, sigma=1.0, shape=(), observed=None):
        return self._add(_RV(name, shape, normal_logp, (as_tensor(mu), as_tensor(sigma)), None, observed))

    def HalfNormal(self, name, sigma=1.0, shape=()):
        return self._add(_RV(name, shape, lambda v, s: halfnormal_logp(v, 0.0, s), (as_tensor(sigma),), "log"))

    def HalfCauchy(self, name, beta=1.0, shape=()):
        return self._add(_RV(name, shape, halfcauchy_logp, (as_tensor(beta),), "log"))

    def Laplace(self, name, mu=0.0, b=1.0, shape=(), observed=None):
        return self._add(_RV(name, shape, laplace_logp, (as_tensor(mu), as_tensor(b)), None, observed))

    def LogNormal(self, name, mu=0.0, sigma=1.0, shape=(), observed=None):
        return self._add(_RV(name, shape, lognormal_logp, (as_tensor(mu), as_tensor(sigma)), "log" if observed is None else None, observed))

    def StudentT(self, name, nu, mu=0.0, sigma=1.0, shape=(), observed=None):
        return self._add(_RV(name, shape, lambda v, m_, s_: studentt_logp(v, as_tensor(float(nu)), m_, s_), (as_tensor(mu), as_tensor(sigma)), None, observed))

    def Beta(self, name, alpha, beta, shape=()):
        return self._add(_RV(name, shape, beta_logp, (as_tensor(float(alpha)), as_tensor(float(beta))), "logodds"))

    def Gamma(self, name, alpha, beta, shape=(), observed=None):
        scale = pt.reciprocal(as_tensor(beta))       # Gamma.dist (continuous.py:2484-2492) hands `scale` to the logp
        return self._add(_RV(name, shape, gamma_logp, (as_tensor(float(alpha)), scale), "log" if observed is None else None, observed))

    def InverseGamma(self, name, alpha, beta, shape=(), observed=None):
        return self._add(_RV(name, shape, invgamma_logp, (as_tensor(float(alpha)), as_tensor(beta)), "log" if observed is None else None, observed))

    def Uniform(self, name, lower=0.0, upper=1.0, shape=()):
        return self._add(_RV(name, shape, uniform_logp, (as_tensor(float(lower)), as_tensor(float(upper))), "interval", bounds=(float(lower), float(upper))))

    def TruncatedNormal(self, name, mu=0.0, sigma=1.0, lower=None, upper=None, shape=(), observed=None):
        lo = None if lower is None else as_tensor(float(lower))
        hi = None if upper is None else as_tensor(float(upper))
        fn = lambda v, m_, s_: truncnormal_logp(v, m_, s_, lo, hi)   # noqa: E731
        if observed is None:   # the reference's default transform of a doubly bounded distribution: interval
            return self._add(_RV(name, shape, fn, (as_tensor(mu), as_tensor(sigma)), "interval", bounds=(float(lower), float(upper))))
        return self._add(_RV(name, shape, fn, (as_tensor(mu), as_tensor(sigma)), None, observed))

    def Binomial(self, name, n, p, observed):
        return self._add(_RV(name, np.shape(observed), binomial_logp, (as_tensor(n), as_tensor(p)), None, observed))

    def Poisson(self, name, mu, observed):
        return self._add(_RV(name, np.shape(observed), poisson_logp, (as_tensor(mu),), None, observed))

    def Bernoulli(self, name, logit_p, observed):
        return self._add(_RV(name, np.shape(observed), bernoulli_logp, (pt.sigmoid(logit_p),), None, observed))   # discrete.py:351-352

    # ---- the model protocol of `lower_to_spec` ----
    @property
    def value_vars(self):
        return [rv.value for rv in self.free]

    @property
    def value_shapes(self):
        return {rv.value.name: rv.shape for rv in self.free}

    @property
    def value_transforms(self):
        code = {"log": 1, "logodds": 2, "interval": 3}
        return {rv.value.name: (code[rv.transform], *(rv.bounds or (0.0, 1.0))) for rv in self.free if rv.transform}

    @property
    def logp_owners(self):
        return [rv.value for rv in self.free] + [None] * len(self.obs)

    @property
    def logp_names(self):
        return [rv.name for rv in self.free + self.obs]

    def logp(self, sum=False):
        out = []
        for rv in self.free + self.obs:
            lp = rv.logp_fn(rv.expr, *rv.params)
            if rv.transform == "log":      # + log|J| = value (LogTransform.log_jac_det, transforms.py:880-891)
                lp = lp + rv.value
            if rv.transform == "interval":   # + log|J| (IntervalTransform.log_jac_det, transforms.py:1055-1070)
                lp = lp + interval_log_jac_det(rv.value, *rv.bounds)
            if rv.transform == "logodds":  # + log|J| = log sigmoid(v) + log1p(-sigmoid(v)) (LogOddsTransform, transforms.py:1076-1088)
                sv = pt.sigmoid(rv.value)
                lp = lp + (pt.log(sv) + pt.log1p(-sv))
            out.append(lp)
        return out
