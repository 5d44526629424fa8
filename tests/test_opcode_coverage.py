"""Every opcode of the expression programs in ONE hand-built model (include/nuts_mi355.h NUTS_E_*; 47 of them).

Why this file exists: fourteen opcodes -- arctan, ceil, cos, digamma, expm1, floor, log10, log1mexp, log2, minimum, neq, not, sin,
tanh -- occur in no spec the device had run when the round's GPU minutes were spent (the lowered reference graphs never needed them;
tanh alone is in a hand-built device test).  The oracle's interpreter is held here to torch autograd of the same expression written
directly in torch (host); the device test holds the engine to the oracle (it first ran on the device in round 5's driver session and
runs in every `-m gpu` session since: `profiles/r06f_lowering_device_tests.log`)."""
import numpy as np
import pytest

from oracle import ref_models
from pymc_amd import model_spec as ms
from pymc_amd.model_spec import ModelBuilder

N = 7
W = np.array([0.3, -1.2, 0.8, 2.0, -0.4, 1.1, -0.7])


def _spec():
    m = ModelBuilder()
    x = m.Normal("x", 0.0, 2.0, shape=(N,))
    s = m.HalfNormal("s", 1.5)
    M = m.math
    w = m.as_expr(W) if hasattr(m, "as_expr") else W
    t1 = M.arctan(x) + M.sin(x) * M.cos(s) + M.expm1(0.3 * x) + M.log2(1.0 + M.sqr(x)) + M.log10(2.0 + M.sqr(x * s))
    t2 = M.digamma(1.5 + M.sqr(x)) + M.log1mexp(-(0.5 + M.sqr(x))) + M.tanh(x * s)
    t3 = M.minimum(x, 0.3 * s + w) + M.maximum(x * 0.5, w) + M.floor(3.0 * x) * 0.01 * x + M.ceil(2.0 * x + w) * 0.01 * s
    t4 = M.switch(M.neq(M.floor(x), 1.0), x, 2.0 * x) + M.switch(M.not_(M.gt(x, 0.2)), M.sqr(x), x)
    t5 = M.switch(M.and_(M.ge(x, -1.0), M.or_(M.lt(x, 0.5), M.le(s, 1.0))), M.abs(x - 0.123), M.sign(x) * 0.5) + M.switch(M.eq(M.ceil(x), 1.0), s, 0.5 * s)
    t6 = M.gammaln(1.2 + M.sqr(x)) + M.erf(x) + M.erfc(x) * 0.1 + M.log(M.erfcx(x)) + M.logaddexp(x, s) + M.clip(x, -0.7, 1.3) + M.pow(1.0 + M.sqr(x), 0.3 * M.tanh(s))
    t7 = M.log1p(M.sqr(x)) + M.sigmoid(x) + M.softplus(x) + M.sqrt(1.0 + M.sqr(x)) + M.reciprocal(2.0 + M.sqr(x)) + M.exp(0.2 * x) + (-x) / (1.0 + M.sqr(s)) + (x - s) ** 3.0 \
        + ms.Expr.op(m, ms.E_SUB, M.exp(0.1 * x), M.exp(0.1 * s))          # (the builder writes `a - b` as an affine term: the opcode by hand)
    m.Potential("every_opcode_a", M.check(t1 + t2 + t3 + t4, M.gt(s, 0.0)))
    m.Potential("every_opcode_b", t5 + t6 + t7)
    return m.build()


def _torch_value(q):
    import torch

    qt = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    x, s = qt[:N], torch.exp(qt[N])
    w = torch.tensor(W)
    ln = torch.log
    sw = torch.where
    t1 = torch.atan(x) + torch.sin(x) * torch.cos(s) + torch.expm1(0.3 * x) + torch.log2(1 + x * x) + torch.log10(2 + (x * s) ** 2)
    a = -(0.5 + x * x)
    t2 = torch.digamma(1.5 + x * x) + ln(-torch.expm1(a)) + torch.tanh(x * s)
    t3 = torch.minimum(x, 0.3 * s + w) + torch.maximum(x * 0.5, w) + torch.floor(3 * x) * 0.01 * x + torch.ceil(2 * x + w) * 0.01 * s
    t4 = sw(torch.floor(x) != 1.0, x, 2 * x) + sw(~(x > 0.2), x * x, x)
    t5 = sw((x >= -1.0) & ((x < 0.5) | (s <= 1.0)), torch.abs(x - 0.123), torch.sign(x) * 0.5) + sw(torch.ceil(x) == 1.0, s, 0.5 * s)
    t6 = torch.lgamma(1.2 + x * x) + torch.erf(x) + torch.erfc(x) * 0.1 + ln(torch.special.erfcx(x)) + torch.logaddexp(x, s.expand_as(x)) \
        + torch.clamp(x, -0.7, 1.3) + (1 + x * x) ** (0.3 * torch.tanh(s))
    t7 = torch.log1p(x * x) + torch.sigmoid(x) + torch.nn.functional.softplus(x) + torch.sqrt(1 + x * x) + 1.0 / (2 + x * x) + torch.exp(0.2 * x) \
        + (-x) / (1 + s * s) + (x - s) ** 3 + (torch.exp(0.1 * x) - torch.exp(0.1 * s))
    half_log_2pi = 0.5 * np.log(2.0 * np.pi)          # (written out: `torch.distributions` would take the scales as float32)
    prior = (-0.5 * (x / 2.0) ** 2 - np.log(2.0) - half_log_2pi).sum() + (-0.5 * (s / 1.5) ** 2 - np.log(1.5) - half_log_2pi + np.log(2.0)) + qt[N]
    lp = prior + (t1 + t2 + t3 + t4 + t5 + t6 + t7).sum()
    lp.backward()
    return float(lp.detach()), qt.grad.numpy().copy()


def _points():
    rng = np.random.default_rng(77)
    return [rng.normal(size=N + 1) * sc for sc in (0.3, 0.6, 1.0, 1.4)]


def test_the_model_uses_every_opcode():
    spec = _spec()
    used = {i.op for f in spec.factors for i in f.prog}
    assert used == set(range(47)), sorted(set(range(47)) - used)
    assert ms.engine_refusal(spec) is None and all(len(f.prog) <= ms.MAX_FACTOR_INSTR for f in spec.factors)


def test_the_oracles_interpreter_agrees_with_torch_autograd_on_every_opcode():
    spec = _spec()
    for q in _points():
        lp0, g0 = _torch_value(q)
        lp, g = ref_models.evaluate(spec, q)
        assert abs(lp - lp0) <= 1e-11 * max(1.0, abs(lp0)), (lp, lp0)
        assert np.max(np.abs(g - g0)) <= 1e-10 * max(1.0, np.max(np.abs(g0))), float(np.max(np.abs(g - g0)))


@pytest.mark.gpu
def test_the_device_agrees_with_the_oracle_on_every_opcode():
    from pymc_amd.value_grad import DeviceValueGradFunction

    spec = _spec()
    f = DeviceValueGradFunction(spec, device=0)
    try:
        for q in _points():
            lp0, g0 = ref_models.evaluate(spec, q)
            lp, g = f._pytensor_function(q)
            assert abs(lp - lp0) <= 1e-9 * max(1.0, abs(lp0)), (lp, lp0)
            assert np.max(np.abs(g - g0)) <= 1e-8 * max(1.0, np.max(np.abs(g0))), float(np.max(np.abs(g - g0)))
    finally:
        f.close()
