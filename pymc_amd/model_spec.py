"""Model specification: the thin IR that is lowered through the C-ABI.

The reference builds the joint log-density as a PyTensor graph
(`Model.logp`, pymc/model/core.py:612-695 ->
`transformed_conditional_logp`, pymc/logprob/basic.py:618-667) and compiles it
with `ValueGradFunction` (pymc/model/core.py:142-305).  PyTensor is not
available on the build image, so the graph walker is a "next" row (SURVEY.md
section 8f-2); what crosses the C-ABI today is this *spec*: a struct-of-arrays
description of

* the free value variables, in `model.value_vars` order, with their default
  transforms (`log`, `logodds`, `interval`; pymc/logprob/transforms.py:880-891,
  967-1088) -- this fixes the layout of the raveled parameter vector exactly as
  `DictToArrayBijection.map` does (pymc/blocking.py:67-75);
* element-wise factors ``dist(value | args)`` whose arguments are affine terms
  ``a + b*c`` over constants, data vectors and variables;
* dense nodes that get their own HBM-streaming kernels: hierarchical
  logistic-regression rows and an MvNormal with constant covariance.

The small builder below mirrors the PyMC model-building idiom
(``m.Normal("mu", 0, 1)``, ``observed=``) so that parity tests read like the
reference's own tests.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

# transform codes (must match include/nuts_mi355.h)
TR_NONE, TR_LOG, TR_LOGODDS, TR_INTERVAL = 0, 1, 2, 3
TRANSFORM_NAMES = {TR_NONE: None, TR_LOG: "log", TR_LOGODDS: "logodds", TR_INTERVAL: "interval"}

# operand kinds
OP_CONST, OP_DATA, OP_VAR, OP_TMP, OP_GATHER = 0, 1, 2, 3, 4   # OP_GATHER: var[idx[i]], `c` = id of the index data vector
OP_LIN = 5   # element i of column int(c) of linear predictor `ref` (ModelSpec.lins): eta = X @ coefficients

# expression-program opcodes (must match include/nuts_mi355.h NUTS_E_*)
(E_ADD, E_SUB, E_MUL, E_DIV, E_NEG, E_EXP, E_LOG, E_LOG1P, E_SIGMOID, E_SOFTPLUS, E_SQRT, E_SQR, E_RECIPROCAL, E_TANH, E_ABS, E_POWC,
 E_GT, E_GE, E_LT, E_LE, E_EQ, E_NEQ, E_AND, E_OR, E_NOT, E_SWITCH, E_GAMMALN, E_ERF, E_ERFC, E_ERFCX, E_LOG1MEXP, E_EXPM1, E_SIGN,
 E_MAXIMUM, E_MINIMUM, E_POW, E_FLOOR, E_CEIL, E_SIN, E_COS, E_ARCTAN, E_LOGADDEXP, E_CLIP, E_CHECK, E_LOG2, E_LOG10, E_DIGAMMA) = range(47)
E_NAMES = {"add": E_ADD, "sub": E_SUB, "mul": E_MUL, "div": E_DIV, "neg": E_NEG, "exp": E_EXP, "log": E_LOG, "log1p": E_LOG1P, "sigmoid": E_SIGMOID,
           "softplus": E_SOFTPLUS, "sqrt": E_SQRT, "sqr": E_SQR, "reciprocal": E_RECIPROCAL, "tanh": E_TANH, "abs": E_ABS, "pow": E_POWC,
           "gt": E_GT, "ge": E_GE, "lt": E_LT, "le": E_LE, "eq": E_EQ, "neq": E_NEQ, "and": E_AND, "or": E_OR, "not": E_NOT, "switch": E_SWITCH,
           "gammaln": E_GAMMALN, "erf": E_ERF, "erfc": E_ERFC, "erfcx": E_ERFCX, "log1mexp": E_LOG1MEXP, "expm1": E_EXPM1, "sign": E_SIGN,
           "maximum": E_MAXIMUM, "minimum": E_MINIMUM, "floor": E_FLOOR, "ceil": E_CEIL, "sin": E_SIN, "cos": E_COS, "arctan": E_ARCTAN,
           "logaddexp": E_LOGADDEXP, "clip": E_CLIP, "check": E_CHECK, "log2": E_LOG2, "log10": E_LOG10, "digamma": E_DIGAMMA}
E_TERNARY = (E_SWITCH, E_CLIP)
E_BINARY = (E_ADD, E_SUB, E_MUL, E_DIV, E_GT, E_GE, E_LT, E_LE, E_EQ, E_NEQ, E_AND, E_OR, E_MAXIMUM, E_MINIMUM, E_POW, E_LOGADDEXP, E_CHECK) + E_TERNARY
MAX_FACTOR_INSTR = 128

# distribution codes (must match include/nuts_mi355.h)
(
    D_NORMAL,
    D_HALFNORMAL,
    D_CAUCHY,
    D_HALFCAUCHY,
    D_STUDENTT,
    D_BETA,
    D_EXPONENTIAL,
    D_UNIFORM,
    D_BERNOULLI_LOGIT,
    D_LOGNORMAL,
    D_BERNOULLI,
    D_TRUNCNORMAL,
    D_POTENTIAL,
    D_BINOMIAL,
    D_GAMMA,
    D_INVGAMMA,
    D_LAPLACE,
    D_POISSON,
    D_DERIVED,
) = range(19)
DIST_NAMES = {
    D_NORMAL: "Normal",
    D_HALFNORMAL: "HalfNormal",
    D_CAUCHY: "Cauchy",
    D_HALFCAUCHY: "HalfCauchy",
    D_STUDENTT: "StudentT",
    D_BETA: "Beta",
    D_EXPONENTIAL: "Exponential",
    D_UNIFORM: "Uniform",
    D_BERNOULLI_LOGIT: "BernoulliLogit",
    D_LOGNORMAL: "LogNormal",
    D_BERNOULLI: "Bernoulli",
    D_TRUNCNORMAL: "TruncatedNormal",
    D_POTENTIAL: "Potential",
    D_BINOMIAL: "Binomial",
    D_GAMMA: "Gamma",
    D_INVGAMMA: "InverseGamma",
    D_LAPLACE: "Laplace",
    D_POISSON: "Poisson",
    D_DERIVED: "Derived",
}


@dataclass(frozen=True)
class Operand:
    kind: int = OP_CONST
    c: float = 0.0
    ref: int = -1  # data id or var id


ZERO = Operand(OP_CONST, 0.0)
ONE = Operand(OP_CONST, 1.0)


@dataclass(frozen=True)
class Term:
    """value = a + b * c (element-wise, size-1 operands broadcast)."""

    a: Operand = ZERO
    b: Operand = ZERO
    c: Operand = ZERO


@dataclass(frozen=True)
class Instr:
    """tmp[i] = op(x, y), element-wise over the factor's elements (size-1 operands broadcast); unary ops ignore y."""

    op: int
    x: Operand = ZERO
    y: Operand = ZERO
    k: float = 0.0
    z: Operand = ZERO       # third operand (E_SWITCH: the `else` branch; E_CLIP: the upper bound)


@dataclass
class FreeVar:
    """One value variable (a slice of the raveled vector)."""

    name: str  # RV name, e.g. "tau"
    shape: Tuple[int, ...]
    transform: int = TR_NONE
    lower: float = 0.0
    upper: float = 1.0
    offset: int = 0
    # the value variable of a simplex-transformed vector (`SimplexTransform`, logprob/transforms.py:1091-1115): K - 1 free elements
    # (`shape`), the variable itself has K.  Element-wise the engine sees an untransformed vector; the node that consumes it (the
    # mixture's Dirichlet weights) owns the transform and its Jacobian.
    simplex: bool = False

    @property
    def size(self) -> int:
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def constrained_shape(self) -> Tuple[int, ...]:
        return tuple(self.shape[:-1]) + (self.shape[-1] + 1,) if self.simplex else tuple(self.shape)

    @property
    def value_name(self) -> str:
        """`{name}_{transform}__` (pymc/util.py:138-155)."""
        t = "simplex" if self.simplex else TRANSFORM_NAMES[self.transform]
        return self.name if t is None else f"{self.name}_{t}__"


@dataclass
class Factor:
    dist: int
    size: int
    args: Tuple[Term, ...]  # args[0] is the value
    konst: float = 0.0  # parameter-only normaliser precomputed on host (lgamma terms)
    name: str = ""
    # expression program (include/nuts_mi355.h, "Expression programs"): what the arguments' OP_TMP operands refer to
    prog: Tuple[Instr, ...] = ()


@dataclass
class LogitRows:
    """y_i ~ Bernoulli(logit_p = x_i . beta_{g(i)}),  beta_g = mu + sigma * z_g.

    Rows must be sorted by group (`group_idx` non-decreasing).
    """

    X: np.ndarray  # [N, D] float64
    y: np.ndarray  # [N] int8
    group_idx: np.ndarray  # [N] int32, sorted
    mu: int  # var id, size D
    sigma: int  # var id, size D (constrained value used)
    z: int  # var id, size G*D
    name: str = "y"


@dataclass
class MvNormalNode:
    """x ~ MvNormal(mu, cov) with constant mu/cov (pymc/distributions/multivariate.py:158-295)."""

    var: int
    mu: np.ndarray
    cov: np.ndarray
    name: str = "x"
    solver: str = "precision"   # "precision": one mat-vec with cov^-1 per leapfrog; "cholesky": y = L^-1 delta, L^-T y (two
                                # mat-vecs with the inverse Cholesky factor: the conditioning of the reference's triangular solves)


@dataclass
class MixtureRows:
    """A Normal mixture over N observed rows with K components, in one of the two forms PyMC writes it:

    * marginal (`assign is None`):  y_i ~ NormalMixture(w, mu, sigma) -- `Mixture` with Normal components,
      logp_i = logsumexp_k(log w_k + logNormal(y_i | mu_k, sigma_k))   (pymc/distributions/mixture.py:469-495);
    * conditional (`assign` = data id of the assignments c):  c_i ~ Categorical(w), y_i ~ Normal(mu[c_i], sigma[c_i]),
      logp_i = log w_{c_i} + logNormal(y_i | mu_{c_i}, sigma_{c_i})    (pymc/distributions/discrete.py:1179-1205,
      continuous.py:526-532); c is an integer-valued entry of `data` that another step method rewrites.

    `mu` is a variable of size K.  `sigma` is a variable of size K (its constrained value is used) or a constant vector.  The
    weights are a constant vector (sum 1), `softmax(logits)` of a variable of size K (`w_logits`), or `w ~ Dirichlet(w_alpha)` under
    PyMC's default simplex transform: then `w_logits` is the value variable `<w>_simplex__` of size K - 1 and the node itself adds
    Dirichlet.logp(w) and the transform's log-Jacobian (`w_alpha` is not None; logprob/transforms.py:1091-1115)."""

    y: np.ndarray                     # [N] float64
    K: int
    mu: int                           # var id
    sigma: Optional[int] = None       # var id, or None with sigma_const
    sigma_const: Optional[np.ndarray] = None
    w_logits: Optional[int] = None    # var id, or None with w_const
    w_const: Optional[np.ndarray] = None
    w_alpha: Optional[np.ndarray] = None   # Dirichlet concentration [K]: `w_logits` is the simplex-transformed value (size K - 1)
    assign: Optional[int] = None      # data id of the assignments (float-coded integers in [0, K)), None: marginal
    name: str = "y"


GLM_NORMAL, GLM_BERNOULLI, GLM_POISSON = 0, 1, 2
GLM_FAMILIES = {"normal": GLM_NORMAL, "bernoulli": GLM_BERNOULLI, "poisson": GLM_POISSON}


@dataclass
class GlmRows:
    """A generalised linear model over N observed rows: eta = intercept + X @ beta (`pm.math.dot`, pymc/math.py:56) as the
    location / logit / log-mean of the likelihood --

        pm.Normal(name, mu=alpha + pm.math.dot(X, beta), sigma=sigma, observed=y)        continuous.py:526-532
        pm.Bernoulli(name, logit_p=alpha + pm.math.dot(X, beta), observed=y)             discrete.py:351-352,362-374
        pm.Poisson(name, mu=pm.math.exp(alpha + pm.math.dot(X, beta)), observed=y)       discrete.py:581-597

    `beta`: variable of size P <= 512 -- or an EXPRESSION of the model's variables (`pm.math.dot(X, mu + sigma * z)`, the
    non-centred hierarchical regression): then `beta` is None and `beta_derived` the index of a D_DERIVED factor of P elements whose
    term is the expression; `intercept`: scalar variable or None; `sigma` (Normal): scalar variable (its constrained value) or None
    with `sigma_const`."""

    X: np.ndarray                     # [N, P] float64
    y: np.ndarray                     # [N] float64
    family: int                       # GLM_*
    beta: Optional[int]               # var id, or None with beta_derived
    intercept: Optional[int] = None   # var id
    sigma: Optional[int] = None       # var id, or None with sigma_const
    sigma_const: float = 1.0
    name: str = "y"
    beta_derived: Optional[int] = None   # index (into ModelSpec.factors) of the D_DERIVED factor that is beta


LIN_MAXK, MAX_LINS, LIN_MAXP = 16, 4, 512


@dataclass
class LinPredictors:
    """K linear predictors that share one constant matrix X [N, P]: eta_k = X @ coef_k (`pm.math.dot(X, beta)`, pymc/math.py:56),
    read by any factor through OP_LIN operands (include/nuts_mi355.h, dense node 5).  `cols[k] = (var, off, stride)`: coefficient p of
    column k is the CONSTRAINED value of element off + p * stride of variable `var` (var >= 0), or that element of the D_DERIVED
    factor -(var + 1).  N == 1: a weighted sum over a long axis (`pt.sum(x)`: X = ones), which broadcasts against any factor."""

    X: np.ndarray
    cols: List[Tuple[int, int, int]] = field(default_factory=list)

    def coef_index(self, spec: "ModelSpec", k: int) -> np.ndarray:
        """Indices of column k's coefficients in the raveled vector (variables only)."""
        var, off, stride = self.cols[k]
        return spec.vars[var].offset + off + stride * np.arange(self.X.shape[1])


@dataclass
class ModelSpec:
    vars: List[FreeVar] = field(default_factory=list)
    data: List[np.ndarray] = field(default_factory=list)
    factors: List[Factor] = field(default_factory=list)
    logit_rows: Optional[LogitRows] = None
    mvnormal: Optional[MvNormalNode] = None
    mixture_rows: Optional[MixtureRows] = None
    glm_rows: Optional[GlmRows] = None
    lins: List[LinPredictors] = field(default_factory=list)
    # "extra" inputs of the log-density (model/core.py:142-190 `extra_vars_and_values`): name -> index into `data`;
    # the caller rewrites them through `set_extra_values` (value variables sampled by another step method)
    extra: Dict[str, int] = field(default_factory=dict)
    # mixture assignments sampled by another step method (pymc_amd/gibbs.py): extras of this spec that are FUNCTIONS of that
    # variable (`mixture.extras_for(point[mixture.name])`) rather than entries of the point
    mixture: Optional[object] = None
    # `pm.Deterministic(name, expr)` (model/core.py:1940-2005): named functions of the variables, evaluated for the trace
    # (backends/base.py:183-191 records them next to the untransformed variables): name -> (program, result term, size)
    deterministics: Dict[str, Tuple[Tuple["Instr", ...], "Term", int]] = field(default_factory=dict)
    # ... and their shapes where they are not vectors (a [N, K] matrix of probabilities): name -> shape; absent: () or (size,)
    deterministic_shapes: Dict[str, Tuple[int, ...]] = field(default_factory=dict)
    # data vectors [n_device_data:] are constants of the Deterministics only (masks, index vectors): the host evaluates those, the
    # device never reads them and they are not uploaded.  None: every vector is the device's
    n_device_data: Optional[int] = None

    @property
    def n(self) -> int:
        return sum(v.size for v in self.vars)

    @property
    def point_map_info(self):
        """`RaveledVars.point_map_info` (pymc/blocking.py:40-46)."""
        return tuple((v.value_name, tuple(v.shape), v.size, np.dtype("float64")) for v in self.vars)


# ---------------------------------------------------------------------------
# PyMC-flavoured builder
# ---------------------------------------------------------------------------


class Expr:
    """An expression over model quantities: an affine term ``a + b*c`` where it can be one (then `term` is it and nothing else is
    stored), otherwise a node ``op(x, y)`` over other expressions (`node`), which `ModelBuilder` turns into the factor's
    expression program (include/nuts_mi355.h)."""

    def __init__(self, builder: "ModelBuilder", term: Optional[Term], size: int, node=None):
        self._b, self._term, self.size, self.node = builder, term, size, node

    @property
    def term(self) -> Term:
        if self.node is not None:
            raise NotImplementedError("expression is outside the affine IR `a + b*c`: it needs an expression program (handled by the "
                                      "distribution / Potential / Deterministic it is handed to)")
        return self._term

    @classmethod
    def op(cls, builder, opcode, x, y=None, k=0.0, z=None):
        x = builder.as_expr(x)
        y = builder.as_expr(y) if y is not None else None
        z = builder.as_expr(z) if z is not None else None
        return cls(builder, None, max(x.size, y.size if y is not None else 1, z.size if z is not None else 1), (opcode, x, y, float(k), z))

    # -- helpers -----------------------------------------------------------
    def _simple(self) -> Optional[Operand]:
        if self.node is not None:
            return None
        t = self._term
        if t.b == ZERO or t.c == ZERO:
            return t.a
        return None

    def _product(self) -> Optional[Tuple[Operand, Operand]]:
        if self.node is not None:
            return None
        t = self._term
        if t.a == ZERO:
            return t.b, t.c
        return None

    def __add__(self, other):
        other = self._b.as_expr(other)
        size = max(self.size, other.size)
        s, o = self._simple(), other._simple()
        if s is not None and o is not None:
            if s.kind == OP_CONST and o.kind == OP_CONST:
                return Expr(self._b, Term(Operand(OP_CONST, s.c + o.c)), size)
            return Expr(self._b, Term(s, o, ONE), size)
        if s is not None and other._product() is not None:
            return Expr(self._b, Term(s, *other._product()), size)
        if o is not None and self._product() is not None:
            return Expr(self._b, Term(o, *self._product()), size)
        return Expr.op(self._b, E_ADD, self, other)

    __radd__ = __add__

    def __mul__(self, other):
        other = self._b.as_expr(other)
        size = max(self.size, other.size)
        s, o = self._simple(), other._simple()
        if s is not None and o is not None:
            if s.kind == OP_CONST and o.kind == OP_CONST:
                return Expr(self._b, Term(Operand(OP_CONST, s.c * o.c)), size)
            return Expr(self._b, Term(ZERO, s, o), size)
        return Expr.op(self._b, E_MUL, self, other)

    __rmul__ = __mul__

    def __sub__(self, other):
        other = self._b.as_expr(other)
        o = other._simple()
        if o is not None and o.kind == OP_CONST:
            return self + (-o.c)
        if o is not None and self._simple() is not None:
            return Expr(self._b, Term(self._simple(), Operand(OP_CONST, -1.0), o), max(self.size, other.size))
        return Expr.op(self._b, E_SUB, self, other)

    def __rsub__(self, other):
        return self._b.as_expr(other) - self

    def __neg__(self):
        s = self._simple()
        if s is not None:
            return Expr(self._b, Term(ZERO, Operand(OP_CONST, -1.0), s), self.size) if s.kind != OP_CONST else Expr(self._b, Term(Operand(OP_CONST, -s.c)), self.size)
        return Expr.op(self._b, E_NEG, self)

    def __truediv__(self, other):
        other = self._b.as_expr(other)
        o = other._simple()
        if o is not None and o.kind == OP_CONST:
            return self * (1.0 / o.c)
        return Expr.op(self._b, E_DIV, self, other)

    def __rtruediv__(self, other):
        return Expr.op(self._b, E_DIV, self._b.as_expr(other), self)

    def __getitem__(self, idx):
        """`a[group_idx]` for a free variable `a`: a gather (varying intercepts / slopes)."""
        s_ = self._simple()
        if s_ is None or s_.kind != OP_VAR:
            raise NotImplementedError("only a free variable itself can be indexed")
        idx = np.asarray(idx)
        if idx.ndim != 1 or idx.dtype.kind not in "iu":
            raise NotImplementedError("index with a one-dimensional integer array")
        n = self._b.spec.vars[s_.ref].size
        if idx.size and (idx.min() < 0 or idx.max() >= n):
            raise IndexError("index out of range for the variable")
        self._b.spec.data.append(np.ascontiguousarray(idx, dtype="float64"))
        return Expr(self._b, Term(Operand(OP_GATHER, float(len(self._b.spec.data) - 1), s_.ref)), idx.size)

    def __pow__(self, k):
        if float(k) == 2.0:
            return Expr.op(self._b, E_SQR, self)
        return Expr.op(self._b, E_POWC, self, None, float(k))


class _Math:
    """`pm.math.*` (pymc/math.py) for builder expressions."""

    def __init__(self, builder):
        self._b = builder

    def _un(self, opcode, x):
        return Expr.op(self._b, opcode, x)

    def exp(self, x): return self._un(E_EXP, x)
    def log(self, x): return self._un(E_LOG, x)
    def log1p(self, x): return self._un(E_LOG1P, x)
    def sigmoid(self, x): return self._un(E_SIGMOID, x)
    invlogit = sigmoid
    def softplus(self, x): return self._un(E_SOFTPLUS, x)
    def sqrt(self, x): return self._un(E_SQRT, x)
    def sqr(self, x): return self._un(E_SQR, x)
    def tanh(self, x): return self._un(E_TANH, x)
    def abs(self, x): return self._un(E_ABS, x)
    def reciprocal(self, x): return self._un(E_RECIPROCAL, x)
    # the scalar ops of the reference's density bodies (include/nuts_mi355.h NUTS_E_*): a density written out op by op
    def gammaln(self, x): return self._un(E_GAMMALN, x)
    def erf(self, x): return self._un(E_ERF, x)
    def erfc(self, x): return self._un(E_ERFC, x)
    def erfcx(self, x): return self._un(E_ERFCX, x)
    def log1mexp(self, x): return self._un(E_LOG1MEXP, x)
    def expm1(self, x): return self._un(E_EXPM1, x)
    def sign(self, x): return self._un(E_SIGN, x)
    def floor(self, x): return self._un(E_FLOOR, x)
    def ceil(self, x): return self._un(E_CEIL, x)
    def sin(self, x): return self._un(E_SIN, x)
    def cos(self, x): return self._un(E_COS, x)
    def arctan(self, x): return self._un(E_ARCTAN, x)
    def digamma(self, x): return self._un(E_DIGAMMA, x)
    def log2(self, x): return self._un(E_LOG2, x)
    def log10(self, x): return self._un(E_LOG10, x)
    def not_(self, x): return self._un(E_NOT, x)
    def gt(self, x, y): return Expr.op(self._b, E_GT, x, y)
    def ge(self, x, y): return Expr.op(self._b, E_GE, x, y)
    def lt(self, x, y): return Expr.op(self._b, E_LT, x, y)
    def le(self, x, y): return Expr.op(self._b, E_LE, x, y)
    def eq(self, x, y): return Expr.op(self._b, E_EQ, x, y)
    def neq(self, x, y): return Expr.op(self._b, E_NEQ, x, y)
    def and_(self, x, y): return Expr.op(self._b, E_AND, x, y)
    def or_(self, x, y): return Expr.op(self._b, E_OR, x, y)
    def maximum(self, x, y): return Expr.op(self._b, E_MAXIMUM, x, y)
    def minimum(self, x, y): return Expr.op(self._b, E_MINIMUM, x, y)
    def pow(self, x, y): return Expr.op(self._b, E_POW, x, y)
    def logaddexp(self, x, y): return Expr.op(self._b, E_LOGADDEXP, x, y)
    def switch(self, c, x, y): return Expr.op(self._b, E_SWITCH, c, x, z=y)
    def clip(self, x, lo, hi): return Expr.op(self._b, E_CLIP, x, lo, z=hi)
    def check(self, x, cond):
        """`check_parameters(x, cond)` (dist_math.py:50-74): x where cond holds for every element of the factor, else the factor is -inf."""
        return Expr.op(self._b, E_CHECK, x, cond)


# the engine's structural limits (csrc/model_dev.h MAX_BTERMS / MAX_FACTOR_BT / MAX_DEFERRED / MAX_DERIVED): what `nuts_model_create`
# checks when it compiles a spec (csrc/engine.hip `compile_spec`)
MAX_BTERMS, MAX_FACTOR_BT, MAX_DEFERRED, MAX_DERIVED = 8, 6, 256, 4


def engine_refusal(spec: "ModelSpec") -> Optional[str]:
    """The reason `nuts_model_create` would refuse the element-wise part of `spec`, or None -- `compile_spec` (csrc/engine.hip)
    restated on the host: operand kinds and references, what broadcasts against what, gather index vectors, the counts the device
    has fixed-size tables for.  A lowering that ends in a refusal can say so before a device is asked; tests hold host-validated
    specs to the limits the device will apply.  (The dense nodes' own shape checks are not restated.)"""
    nv = len(spec.vars)
    data = spec.data if getattr(spec, "n_device_data", None) is None else spec.data[: spec.n_device_data]
    nd = len(data)
    deferred = sum(v.size for v in spec.vars if v.size == 1)
    rows = spec.logit_rows
    if rows is not None:
        deferred += sum(spec.vars[k].size for k in (rows.mu, rows.sigma) if spec.vars[k].size != 1)
    bterm_vars: List[int] = []
    n_derived = 0
    for f in spec.factors:
        prog = tuple(getattr(f, "prog", ()) or ())
        if not 1 <= len(f.args) <= 4 or f.size < 1:
            return "factor with a bad argument count or size"
        if f.dist == D_DERIVED:
            n_derived += 1
            if n_derived > MAX_DERIVED:
                return "too many derived vectors (MAX_DERIVED)"
        if len(prog) > MAX_FACTOR_INSTR:
            return "factor with a bad expression program (offset / length)"
        ops = []
        for i, ins in enumerate(prog):
            used = [ins.x] + ([ins.y] if ins.op in E_BINARY else []) + ([ins.z] if ins.op in E_TERNARY else [])
            if any(o.kind == OP_TMP and not 0 <= o.ref < i for o in used):
                return "expression program: an instruction may only use the results of earlier instructions"
            ops += used
        for t in f.args:
            for o in (t.a, t.b, t.c):
                if o.kind == OP_TMP and not 0 <= o.ref < len(prog):
                    return "factor argument refers to a missing instruction" if prog else "factor argument refers to an instruction but the factor has no program"
                ops.append(o)
        n_bt, seen = 0, set()
        for o in ops:
            if o.kind == OP_GATHER:
                did = int(o.c)
                if not 0 <= o.ref < nv:
                    return "gather refers to a missing variable"
                if not 0 <= did < nd or float(did) != float(o.c):
                    return "gather refers to a missing index vector"
                idx = np.asarray(data[did])
                if idx.size != f.size:
                    return "gather: one index per element of the factor"
                if np.any(idx != np.floor(idx)) or np.any(idx < 0) or np.any(idx >= spec.vars[o.ref].size):
                    return "gather index out of range for its variable"
            elif o.kind == OP_LIN:
                lins = getattr(spec, "lins", [])
                if not 0 <= o.ref < len(lins):
                    return "operand refers to a missing linear predictor"
                if not 0 <= int(o.c) < len(lins[o.ref].cols) or float(int(o.c)) != float(o.c):
                    return "operand refers to a missing column of a linear predictor"
                if lins[o.ref].X.shape[0] not in (1, f.size):
                    return "linear predictor: one row per element of the factor (or a single row that broadcasts)"
                if f.dist == D_DERIVED:
                    return "a derived vector cannot read a linear predictor"
            elif o.kind == OP_DATA:
                if not 0 <= o.ref < nd:
                    return "factor refers to a missing data vector"
                if np.asarray(data[o.ref]).size not in (1, f.size):
                    return "data vector does not broadcast against its factor"
            elif o.kind == OP_VAR:
                if not 0 <= o.ref < nv:
                    return "factor refers to a missing variable"
                if prog and o.ref in seen:          # (a program's variables are registered once each; plain arguments per occurrence)
                    continue
                seen.add(o.ref)
                vs = spec.vars[o.ref].size
                if vs == f.size:
                    continue
                if vs != 1:
                    return "variable does not broadcast against its factor"
                if o.ref not in bterm_vars:
                    if len(bterm_vars) >= MAX_BTERMS:
                        return "too many scalar variables broadcast against vector factors (MAX_BTERMS)"
                    bterm_vars.append(o.ref)
                n_bt += 1
                if n_bt > MAX_FACTOR_BT:
                    return "too many scalar operands in one factor (MAX_FACTOR_BT)"
    if deferred > MAX_DEFERRED:
        return "too many scalar / hyper-parameter elements (MAX_DEFERRED)"
    if getattr(spec, "glm_rows", None) is not None and spec.mvnormal is not None and spec.mvnormal.solver != "precision":
        return "GLM node next to an MvNormal node: the MvNormal node's precision solver only"
    lins = getattr(spec, "lins", [])
    if len(lins) > MAX_LINS:
        return "bad number of linear predictors (NUTS_MAX_LINS)"
    if lins and (spec.logit_rows is not None or getattr(spec, "mixture_rows", None) is not None or getattr(spec, "glm_rows", None) is not None):
        return "linear predictors are not combined with the logit-rows, mixture or GLM node"
    for L in lins:
        N, P = L.X.shape
        K = len(L.cols)
        if N < 1 or P < 1 or not 1 <= K <= LIN_MAXK:
            return "linear predictor with bad dimensions"
        if N > 1 and (P > LIN_MAXP or K * P > 4096):
            return "linear predictor: P <= 512 and K P <= 4096 for predictors with more than one row"
        for var, off, stride in L.cols:
            size = spec.vars[var].size if 0 <= var < nv else (spec.factors[-(var + 1)].size if var < 0 and -(var + 1) < len(spec.factors) and spec.factors[-(var + 1)].dist == D_DERIVED else -1)
            if size < 0:
                return "linear predictor: coefficients must be a variable or a NUTS_D_DERIVED factor"
            if off < 0 or stride < 0 or (stride == 0 and P > 1) or off + (P - 1) * stride >= size:
                return "linear predictor: coefficients beyond the end of their variable"
    return None


def eval_program(spec: "ModelSpec", prog, term: Term, x: np.ndarray) -> np.ndarray:
    """Value of `term` over the expression program `prog` at the CONSTRAINED values `x` (raveled, spec layout).  Host arithmetic
    for the trace's Deterministics only -- the log-density's programs are interpreted on the device."""
    tmp: List[np.ndarray] = []

    def val(o: Operand):
        if o.kind == OP_CONST:
            return np.asarray(o.c)
        if o.kind == OP_DATA:
            d = spec.data[o.ref]
            return d if d.size > 1 else d.reshape(())
        if o.kind == OP_TMP:
            return tmp[o.ref]
        if o.kind == OP_LIN:
            L = spec.lins[o.ref]
            if L.cols[int(o.c)][0] < 0:
                raise NotImplementedError("a Deterministic over a linear predictor whose coefficients are a derived vector")
            eta = x[..., L.coef_index(spec, int(o.c))] @ L.X.T
            return eta if L.X.shape[0] > 1 else (eta[..., 0:1] if x.ndim > 1 else eta.reshape(()))
        v = spec.vars[o.ref]
        if o.kind == OP_GATHER:
            return x[..., v.offset + spec.data[int(o.c)].astype(np.int64)]
        blk = x[..., v.offset : v.offset + v.size]
        return blk if v.size > 1 else blk[..., 0:1] if x.ndim > 1 else blk.reshape(())

    from scipy import special as _sp

    un = {E_NEG: np.negative, E_EXP: np.exp, E_LOG: np.log, E_LOG1P: np.log1p, E_SIGMOID: lambda a: 1.0 / (1.0 + np.exp(-a)),
          E_SOFTPLUS: lambda a: np.logaddexp(0.0, a), E_SQRT: np.sqrt, E_SQR: np.square, E_RECIPROCAL: np.reciprocal, E_TANH: np.tanh, E_ABS: np.abs,
          E_NOT: lambda a: (np.asarray(a) == 0).astype("float64"), E_GAMMALN: _sp.gammaln, E_ERF: _sp.erf, E_ERFC: _sp.erfc, E_ERFCX: _sp.erfcx,
          E_LOG1MEXP: lambda a: np.where(a > -math.log(2.0), np.log(-np.expm1(a)), np.log1p(-np.exp(a))), E_EXPM1: np.expm1, E_SIGN: np.sign,
          E_FLOOR: np.floor, E_CEIL: np.ceil, E_SIN: np.sin, E_COS: np.cos, E_ARCTAN: np.arctan, E_LOG2: np.log2, E_LOG10: np.log10, E_DIGAMMA: _sp.digamma}
    f64 = lambda fn: (lambda a, b: np.asarray(fn(a, b), dtype="float64"))   # noqa: E731
    bi = {E_ADD: np.add, E_SUB: np.subtract, E_MUL: np.multiply, E_DIV: np.divide, E_GT: f64(np.greater), E_GE: f64(np.greater_equal), E_LT: f64(np.less),
          E_LE: f64(np.less_equal), E_EQ: f64(np.equal), E_NEQ: f64(np.not_equal), E_AND: f64(lambda a, b: (np.asarray(a) != 0) & (np.asarray(b) != 0)),
          E_OR: f64(lambda a, b: (np.asarray(a) != 0) | (np.asarray(b) != 0)), E_MAXIMUM: np.maximum, E_MINIMUM: np.minimum, E_POW: np.power,
          E_LOGADDEXP: np.logaddexp, E_CHECK: lambda a, b: np.where(np.asarray(b) != 0, a, np.nan)}
    with np.errstate(all="ignore"):
        for ins in prog:
            if ins.op in bi:
                tmp.append(bi[ins.op](val(ins.x), val(ins.y)))
            elif ins.op == E_SWITCH:
                tmp.append(np.where(np.asarray(val(ins.x)) != 0, val(ins.y), val(ins.z)))
            elif ins.op == E_CLIP:
                tmp.append(np.minimum(np.maximum(val(ins.x), val(ins.y)), val(ins.z)))
            elif ins.op == E_POWC:
                tmp.append(np.power(val(ins.x), ins.k))
            else:
                tmp.append(un[ins.op](val(ins.x)))
        return np.asarray(val(term.a) + val(term.b) * val(term.c), dtype="float64")


_DEFAULT_TRANSFORM = {
    D_HALFNORMAL: TR_LOG,
    D_HALFCAUCHY: TR_LOG,
    D_EXPONENTIAL: TR_LOG,
    D_LOGNORMAL: TR_LOG,
    D_BETA: TR_LOGODDS,
    D_UNIFORM: TR_INTERVAL,
    D_TRUNCNORMAL: TR_INTERVAL,
    D_GAMMA: TR_LOG,
    D_INVGAMMA: TR_LOG,
}


class ModelBuilder:
    """Tiny stand-in for ``pm.Model`` that emits a :class:`ModelSpec`."""

    def __init__(self):
        self.spec = ModelSpec()
        self._names: Dict[str, Expr] = {}
        self.math = _Math(self)

    # -- expression programs -------------------------------------------------
    def _lower_args(self, exprs: List[Expr]):
        """[Expr] -> (terms, program): affine expressions stay plain terms; the others become instructions of ONE program shared by
        the factor's arguments (common sub-expressions, by identity, are emitted once)."""
        prog: List[Instr] = []
        memo: Dict[int, Operand] = {}

        def emit(op, x, y=ZERO, k=0.0, z=ZERO) -> Operand:
            prog.append(Instr(op, x, y, k, z))
            if len(prog) > MAX_FACTOR_INSTR:
                raise NotImplementedError(f"expression needs more than {MAX_FACTOR_INSTR} instructions in one factor")
            return Operand(OP_TMP, 0.0, len(prog) - 1)

        def operand(e: Expr) -> Operand:
            if e.node is None:
                s_ = e._simple()
                if s_ is not None:
                    return s_
                t = e._term
                if t.c == ONE or t.b == ONE:           # a + b: one instruction
                    return emit(E_ADD, t.a, t.b if t.c == ONE else t.c)
                prod = emit(E_MUL, t.b, t.c)
                return prod if t.a == ZERO else emit(E_ADD, t.a, prod)
            if id(e) in memo:
                return memo[id(e)]
            op, x, y, k, z = e.node
            out = emit(op, operand(x), operand(y) if y is not None else ZERO, k, operand(z) if z is not None else ZERO)
            memo[id(e)] = out
            return out

        terms = [e._term if e.node is None else Term(operand(e)) for e in exprs]
        return tuple(terms), tuple(prog)

    def Deterministic(self, name, expr):
        """`pm.Deterministic(name, expr)` (model/core.py:1940-2005): recorded in the trace, no contribution to the log-density."""
        e = self.as_expr(expr)
        (term,), prog = self._lower_args([e])
        self.spec.deterministics[name] = (prog, term, e.size)
        return e

    # -- operands ----------------------------------------------------------
    def as_expr(self, x) -> Expr:
        if isinstance(x, Expr):
            return x
        arr = np.asarray(x, dtype="float64")
        if arr.ndim == 0 or arr.size == 1:
            return Expr(self, Term(Operand(OP_CONST, float(arr.reshape(-1)[0]))), 1)
        self.spec.data.append(np.ascontiguousarray(arr.ravel()))
        return Expr(self, Term(Operand(OP_DATA, 0.0, len(self.spec.data) - 1)), arr.size)

    def dot(self, X, beta):
        """`pm.math.dot(X, beta)` (pymc/math.py:56) for a constant matrix X [N, P] and a free variable `beta` of shape (P,) -- one
        predictor, an expression of N elements -- or (P, K): the K columns of `X @ beta` as a list of expressions.  Usable inside
        any argument of any factor (dense node 5, include/nuts_mi355.h)."""
        X = np.ascontiguousarray(np.asarray(X, dtype="float64"))
        if X.ndim == 1:
            X = X[None, :]
        beta = self.as_expr(beta)
        N, P = X.shape
        s_ = beta._simple()
        if s_ is None or s_.kind != OP_VAR:    # coefficients that are an expression of the variables: a derived vector (D_DERIVED)
            if beta.size != P:
                raise ValueError(f"dot: X is {X.shape}, beta has {beta.size} elements")
            terms, prog = self._lower_args([beta])
            self.spec.factors.append(Factor(D_DERIVED, P, terms, 0.0, f"lin{len(self.spec.lins)}_coef", prog))
            self.spec.lins.append(LinPredictors(X, [(-(len(self.spec.factors) - 1) - 1, 0, 1)]))
            return Expr(self, Term(Operand(OP_LIN, 0.0, len(self.spec.lins) - 1)), N)
        v = self.spec.vars[s_.ref]
        shp = tuple(v.shape) or (1,)
        if shp[0] != P or len(shp) > 2:
            raise ValueError(f"dot: X is {X.shape}, beta is {shp}")
        K = shp[1] if len(shp) == 2 else 1
        cols = [(s_.ref, k, K) for k in range(K)] if len(shp) == 2 else [(s_.ref, 0, 1)]
        self.spec.lins.append(LinPredictors(X, cols))
        lid = len(self.spec.lins) - 1
        out = [Expr(self, Term(Operand(OP_LIN, float(k), lid)), N) for k in range(K)]
        return out if len(shp) == 2 else out[0]

    def sum(self, x):
        """`pt.sum(x)` of a free vector variable over its (long) axis: a one-row predictor with X = ones."""
        e = self.as_expr(x)
        return self.dot(np.ones((1, e.size)), e)

    def Extra(self, name, value) -> Expr:
        """A non-gradient input of the log-density (an `extra_var` of `ValueGradFunction`, model/core.py:142-190):
        kept as a data vector that `set_extra_values({name: ...})` rewrites."""
        arr = np.ascontiguousarray(np.asarray(value, dtype="float64").ravel()).copy()
        self.spec.data.append(arr)
        self.spec.extra[name] = len(self.spec.data) - 1
        return Expr(self, Term(Operand(OP_DATA, 0.0, len(self.spec.data) - 1)), arr.size)

    def Potential(self, name, expr):
        """`pm.Potential(name, expr)`: adds sum(expr) to the joint log-density (model/core.py:666-695)."""
        e = self.as_expr(expr)
        terms, prog = self._lower_args([e])
        self.spec.factors.append(Factor(D_POTENTIAL, e.size, terms, 0.0, name, prog))
        return None

    def _register(self, dist, name, params, shape, observed, transform, bounds=(0.0, 1.0), konst=0.0):
        params = [self.as_expr(p) for p in params]
        if observed is not None:
            val = self.as_expr(np.asarray(observed, dtype="float64"))
            if val.term.a.kind == OP_CONST:  # scalar observation: keep as data of size 1
                self.spec.data.append(np.array([val.term.a.c]))
                val = Expr(self, Term(Operand(OP_DATA, 0.0, len(self.spec.data) - 1)), 1)
            size = max([val.size] + [p.size for p in params])
            terms, prog = self._lower_args([val, *params])
            self.spec.factors.append(Factor(dist, size, terms, konst, name, prog))
            return None
        if shape is None:
            shape = ()
        elif isinstance(shape, int):
            shape = (shape,)
        tr = _DEFAULT_TRANSFORM.get(dist, TR_NONE) if transform == "default" else (transform or TR_NONE)
        var = FreeVar(name, tuple(shape), tr, float(bounds[0]), float(bounds[1]), self.spec.n)
        self.spec.vars.append(var)
        vid = len(self.spec.vars) - 1
        e = Expr(self, Term(Operand(OP_VAR, 0.0, vid)), var.size)
        for p in params:
            if p.size not in (1, var.size):
                raise ValueError(f"parameter of size {p.size} does not broadcast to {name} of size {var.size}")
        terms, prog = self._lower_args([e, *params])
        self.spec.factors.append(Factor(dist, var.size, terms, konst, name, prog))
        self._names[name] = e
        return e

    def Flat(self, name, shape=None):
        """`pm.Flat` (pymc/distributions/continuous.py:351-385): improper flat prior, logp = 0 -- a free value variable
        without a factor of its own (what the raw `pt.vector` inputs of tests/model/test_core.py:318-402 amount to)."""
        if shape is None:
            shape = ()
        elif isinstance(shape, int):
            shape = (shape,)
        var = FreeVar(name, tuple(shape), TR_NONE, 0.0, 1.0, self.spec.n)
        self.spec.vars.append(var)
        e = Expr(self, Term(Operand(OP_VAR, 0.0, len(self.spec.vars) - 1)), var.size)
        self._names[name] = e
        return e

    def Dirichlet(self, name, a):
        """`pm.Dirichlet(name, a=a)` (pymc/distributions/multivariate.py `Dirichlet`) as the weights of a mixture: the value
        variable is `<name>_simplex__` with K - 1 elements (the default transform, distributions/transforms.py `simplex`); hand the
        result to `NormalMixture(w=...)`, which owns the prior's log-density and the transform's Jacobian (K >= 3)."""
        a = np.ascontiguousarray(a, dtype="float64").ravel()
        if a.size < 3 or np.any(a <= 0):
            raise ValueError("Dirichlet: K >= 3 concentrations, all > 0")     # multivariate.py: check_parameters(a > 0)
        var = FreeVar(name, (a.size - 1,), TR_NONE, 0.0, 1.0, self.spec.n, simplex=True)
        self.spec.vars.append(var)
        e = Expr(self, Term(Operand(OP_VAR, 0.0, len(self.spec.vars) - 1)), var.size)
        e.dirichlet_alpha = a
        self._names[name] = e
        return e

    # -- distributions (signatures follow pymc/distributions/continuous.py) --
    def Normal(self, name, mu=0.0, sigma=1.0, shape=None, observed=None):
        return self._register(D_NORMAL, name, (mu, sigma), shape, observed, TR_NONE)

    def HalfNormal(self, name, sigma=1.0, shape=None, observed=None, transform="default"):
        return self._register(D_HALFNORMAL, name, (sigma,), shape, observed, transform)

    def Cauchy(self, name, alpha=0.0, beta=1.0, shape=None, observed=None):
        return self._register(D_CAUCHY, name, (alpha, beta), shape, observed, TR_NONE)

    def HalfCauchy(self, name, beta=1.0, shape=None, observed=None, transform="default"):
        return self._register(D_HALFCAUCHY, name, (beta,), shape, observed, transform)

    def Exponential(self, name, lam=1.0, shape=None, observed=None, transform="default"):
        return self._register(D_EXPONENTIAL, name, (lam,), shape, observed, transform)

    def StudentT(self, name, nu, mu=0.0, sigma=1.0, shape=None, observed=None):
        nu = float(nu)  # constant-only: keeps digamma out of the device gradient
        konst = math.lgamma((nu + 1.0) / 2.0) - math.lgamma(nu / 2.0) - 0.5 * math.log(nu * math.pi)
        return self._register(D_STUDENTT, name, (nu, mu, sigma), shape, observed, TR_NONE, konst=konst)

    def Beta(self, name, alpha, beta, shape=None, observed=None, transform="default"):
        alpha, beta = float(alpha), float(beta)
        konst = -(math.lgamma(alpha) + math.lgamma(beta) - math.lgamma(alpha + beta))
        return self._register(D_BETA, name, (alpha, beta), shape, observed, transform, konst=konst)

    def Uniform(self, name, lower=0.0, upper=1.0, shape=None, observed=None, transform="default"):
        lower, upper = float(lower), float(upper)
        return self._register(D_UNIFORM, name, (lower, upper), shape, observed, transform, bounds=(lower, upper))

    def TruncatedNormal(self, name, mu=0.0, sigma=1.0, lower=None, upper=None, shape=None, observed=None, transform="default"):
        """`pm.TruncatedNormal` (pymc/distributions/continuous.py:596-746) with constant bounds; a free variable needs
        both bounds (interval transform, the reference's default for a doubly bounded distribution)."""
        lo = -math.inf if lower is None else float(lower)
        hi = math.inf if upper is None else float(upper)
        if observed is None and not (math.isfinite(lo) and math.isfinite(hi)):
            raise NotImplementedError("a free TruncatedNormal needs both bounds here (interval transform)")
        bounds = (lo, hi) if observed is None else (0.0, 1.0)
        return self._register(D_TRUNCNORMAL, name, (mu, sigma, lo), shape, observed, transform, konst=hi, bounds=bounds)

    def LogNormal(self, name, mu=0.0, sigma=1.0, shape=None, observed=None, transform="default"):
        return self._register(D_LOGNORMAL, name, (mu, sigma), shape, observed, transform)

    def Bernoulli(self, name, p, observed):
        """`pm.Bernoulli(name, p=..., observed=...)` (pymc/distributions/discrete.py:362-374)."""
        return self._register(D_BERNOULLI, name, (p,), None, observed, TR_NONE)

    def Gamma(self, name, alpha, beta, shape=None, observed=None, transform="default"):
        """`pm.Gamma(name, alpha, beta)` (continuous.py:2400-2521); alpha constant (keeps digamma out of the gradient)."""
        alpha = float(alpha)
        return self._register(D_GAMMA, name, (alpha, beta), shape, observed, transform, konst=-math.lgamma(alpha))

    def InverseGamma(self, name, alpha, beta, shape=None, observed=None, transform="default"):
        """`pm.InverseGamma(name, alpha, beta)` (continuous.py:2540-2639); alpha constant."""
        alpha = float(alpha)
        return self._register(D_INVGAMMA, name, (alpha, beta), shape, observed, transform, konst=-math.lgamma(alpha))

    def Laplace(self, name, mu=0.0, b=1.0, shape=None, observed=None):
        return self._register(D_LAPLACE, name, (mu, b), shape, observed, TR_NONE)

    def Poisson(self, name, mu, observed):
        """`pm.Poisson(name, mu=..., observed=...)` (discrete.py:520-597); `factln(y)` depends on data only."""
        from scipy.special import gammaln

        y = np.asarray(observed, dtype="float64")
        return self._register(D_POISSON, name, (mu, gammaln(y + 1)), None, y, TR_NONE)

    def Binomial(self, name, n, p, observed):
        """`pm.Binomial(name, n=..., p=..., observed=...)` (pymc/distributions/discrete.py:60-154); `binomln(n, y)`
        (dist_math.py:109-114) depends on data only and is taken here."""
        from scipy.special import gammaln

        y = np.asarray(observed, dtype="float64")
        nn = np.broadcast_to(np.asarray(n, dtype="float64"), y.shape)
        lbc = gammaln(nn + 1) - gammaln(y + 1) - gammaln(nn - y + 1)
        n_arg = float(np.asarray(n)) if np.ndim(n) == 0 else nn   # a scalar n stays a constant of the factor
        return self._register(D_BINOMIAL, name, (n_arg, p, lbc), None, y, TR_NONE)

    def BernoulliLogit(self, name, logit_p, observed):
        """`pm.Bernoulli(name, logit_p=..., observed=...)` (pymc/distributions/discrete.py:343-374)."""
        return self._register(D_BERNOULLI_LOGIT, name, (logit_p,), None, observed, TR_NONE)

    # -- dense nodes ---------------------------------------------------------
    def _var_id(self, e: Expr) -> int:
        op = e._simple()
        if op is None or op.kind != OP_VAR:
            raise ValueError("expected a free variable")
        return op.ref

    def HierLogitRows(self, name, X, y, group_idx, mu: Expr, sigma: Expr, z: Expr):
        X = np.ascontiguousarray(X, dtype="float64")
        g = np.ascontiguousarray(group_idx, dtype="int32")
        if np.any(np.diff(g) < 0):
            order = np.argsort(g, kind="stable")
            X, g, y = X[order], g[order], np.asarray(y)[order]
        self.spec.logit_rows = LogitRows(
            X, np.ascontiguousarray(y, dtype="int8"), g, self._var_id(mu), self._var_id(sigma), self._var_id(z), name
        )

    def GLM(self, name, X, beta: Expr, observed, family="normal", intercept: Optional[Expr] = None, sigma=1.0):
        """The likelihood of a generalised linear model (`GlmRows`): `family` "normal" (with `sigma` a scalar variable or a
        constant), "bernoulli" (logit link) or "poisson" (log link); eta = intercept + X @ beta."""
        X = np.ascontiguousarray(X, dtype="float64")
        y = np.ascontiguousarray(observed, dtype="float64").ravel()
        beta = self.as_expr(beta)
        if X.ndim != 2 or X.shape[0] != y.size or X.shape[1] != beta.size:
            raise ValueError("GLM: X is [N, P], beta has P elements, one observation per row")
        bo = beta._simple()
        if bo is not None and bo.kind == OP_VAR and self.spec.vars[bo.ref].transform == TR_NONE:
            node = GlmRows(X, y, GLM_FAMILIES[family], bo.ref, name=name)
        else:    # beta an expression: a derived vector the node reads (include/nuts_mi355.h NUTS_D_DERIVED)
            terms, prog = self._lower_args([beta])
            self.spec.factors.append(Factor(D_DERIVED, beta.size, terms, 0.0, name + "_beta", prog))
            node = GlmRows(X, y, GLM_FAMILIES[family], None, name=name, beta_derived=len(self.spec.factors) - 1)
        if intercept is not None:
            node.intercept = self._var_id(intercept)
        if isinstance(sigma, Expr):
            node.sigma = self._var_id(sigma)
        else:
            node.sigma_const = float(sigma)
        self.spec.glm_rows = node

    def MvNormal(self, name, mu, cov, solver="precision"):
        if solver not in ("precision", "cholesky"):
            raise ValueError("solver must be 'precision' or 'cholesky'")
        mu = np.ascontiguousarray(mu, dtype="float64")
        cov = np.ascontiguousarray(cov, dtype="float64")
        var = FreeVar(name, (len(mu),), TR_NONE, 0.0, 1.0, self.spec.n)
        self.spec.vars.append(var)
        vid = len(self.spec.vars) - 1
        self.spec.mvnormal = MvNormalNode(vid, mu, cov, name, solver)
        e = Expr(self, Term(Operand(OP_VAR, 0.0, vid)), var.size)
        self._names[name] = e
        return e

    def NormalMixture(self, name, w, mu: Expr, sigma, observed, assign: Optional[Expr] = None):
        """`pm.NormalMixture(name, w=w, mu=mu, sigma=sigma, observed=y)` (marginal form), or with `assign` = an `Extra` holding the
        assignments c: `pm.Categorical("c", p=w, shape=N)` + `pm.Normal(name, mu[c], sigma[c], observed=y)` evaluated for the
        continuous variables.  `w`: a constant vector, or `m.math.softmax(logits)` written as `("softmax", logits_expr)`;
        `sigma`: a variable of size K or a constant (scalar / vector)."""
        y = np.ascontiguousarray(observed, dtype="float64").ravel()
        K = mu.size
        node = MixtureRows(y, K, self._var_id(mu), name=name)
        if isinstance(sigma, Expr):
            node.sigma = self._var_id(sigma)
        else:
            node.sigma_const = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, dtype="float64"), (K,)))
        if isinstance(w, tuple) and len(w) == 2 and w[0] == "softmax":
            node.w_logits = self._var_id(w[1])
        elif isinstance(w, Expr) and getattr(w, "dirichlet_alpha", None) is not None:
            node.w_logits = self._var_id(w)
            node.w_alpha = np.ascontiguousarray(w.dirichlet_alpha, dtype="float64")
            if node.w_alpha.shape != (K,) or K < 3:
                raise ValueError("Dirichlet mixture weights: K >= 3 components, one concentration per component")
        else:
            wc = np.ascontiguousarray(w, dtype="float64")
            if wc.shape != (K,) or np.any(wc < 0) or not np.isclose(wc.sum(), 1.0):
                raise ValueError("constant mixture weights: a vector of K non-negative numbers that sum to 1")
            node.w_const = wc
        if assign is not None:
            t = assign.term
            if t.a.kind != OP_DATA or t.b.kind != OP_CONST or t.b.c != 0.0:
                raise ValueError("assign must be an Extra / Data entry holding the assignments")
            node.assign = int(t.a.ref)
            if self.spec.data[node.assign].size != y.size:
                raise ValueError("one assignment per observed row")
        for vid in (node.mu, node.sigma, node.w_logits):
            want = K - 1 if (vid == node.w_logits and node.w_alpha is not None) else K
            if vid is not None and self.spec.vars[vid].size != want:
                raise ValueError("mixture parameters must have K elements (K - 1 for simplex-transformed Dirichlet weights)")
        self.spec.mixture_rows = node

    def build(self) -> ModelSpec:
        return self.spec
