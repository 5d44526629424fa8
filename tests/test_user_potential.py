"""A potential the caller owns: `NUTS(potential=<user subclass of QuadPotential>)`.

The reference's contract is `test_user_potential` (tests/step_methods/hmc/test_quadpotential.py:138-158): a subclass may
override `velocity` / `energy` / `velocity_energy` / `random` (and `update`, `reset`, `raise_ok`, `stats`) and the sampler
calls ITS methods.  Here such a potential makes the step method create a NUTS_POT_HOST chain (include/nuts_mi355.h): logp and
gradient, kicks, tree and acceptance stay on the device, the potential's methods are called back where the reference's
integrator calls them (integration.py:72-73,121,134).

CPU part: which classes are routed that way, and the host arithmetic `super()` reaches.  GPU part: the callbacks really are
called at the reference's call sites, a wholly user-defined potential samples like the oracle driven by the same object, and
an exception raised inside a callback comes out of `astep`.
"""

import collections

import numpy as np
import pytest
import scipy.linalg

from oracle import ref_models, ref_sampler
from pymc_amd import models
from pymc_amd.quadpotential import (
    QuadPotential, QuadPotentialDiag, QuadPotentialDiagAdapt, QuadPotentialFull, QuadPotentialFullInv, _user_overrides,
)
from pymc_amd.step import _host_potential_wanted

INT_KEYS = ("depth", "tree_size", "index_in_trajectory", "diverging")


class CountingDiag(QuadPotentialDiag):
    """The reference test's potential: QuadPotentialDiag with `energy` overridden."""

    def __init__(self, v):
        super().__init__(v)
        self.called = collections.Counter()

    def energy(self, x, velocity=None):
        self.called["energy"] += 1
        return super().energy(x, velocity)


class LowRankPotential(QuadPotential):
    """Nothing of the library in it: covariance (inverse mass matrix) D + U U^T, applied without forming the matrix."""

    def __init__(self, d, U, rng=None):
        super().__init__(rng)
        self.d, self.U = np.asarray(d, float), np.asarray(U, float)
        # p ~ N(0, M), M = (D + U U^T)^-1  <=>  p = L^-T z with L L^T = D + U U^T
        self.L = scipy.linalg.cholesky(np.diag(self.d) + self.U @ self.U.T, lower=True)
        self.called = collections.Counter()
        self.seen = []

    def velocity(self, x, out=None):
        self.called["velocity"] += 1
        v = self.d * x + self.U @ (self.U.T @ x)
        if out is None:
            return v
        out[:] = v
        return None

    def energy(self, x, velocity=None):
        self.called["energy"] += 1
        return 0.5 * np.dot(x, self.velocity(x) if velocity is None else velocity)

    def velocity_energy(self, x, v_out):
        self.called["velocity_energy"] += 1
        v_out[:] = self.d * x + self.U @ (self.U.T @ x)
        return 0.5 * np.dot(x, v_out)

    def random(self):
        self.called["random"] += 1
        return scipy.linalg.solve_triangular(self.L.T, self.rng.normal(size=len(self.d)), lower=False)

    def update(self, sample, grad, tune):
        self.called["update"] += 1
        self.seen.append((np.array(sample), bool(tune)))

    def stats(self):
        return {"largest_eigval": 2.5, "smallest_eigval": 0.5}


class _OracleView(ref_sampler.PotentialBase):
    """The same object behind the oracle's potential interface (`velocity(p)` returns the array)."""

    def __init__(self, pot):
        self.pot = pot

    rng = property(lambda self: self.pot.rng, lambda self, r: setattr(self.pot, "rng", r))

    def velocity(self, p):
        return self.pot.velocity(p)

    def random(self):
        return self.pot.random()


# ---- CPU: routing and host arithmetic ---------------------------------------------------------------------------------------

def test_which_potentials_are_called_back():
    v = np.array([1.0, 2.0, 4.0])
    assert _user_overrides(QuadPotentialDiag(v)) == []
    assert not _host_potential_wanted(QuadPotentialDiag(v))
    assert not _host_potential_wanted(QuadPotentialFull(np.diag(v)))
    assert not _host_potential_wanted(QuadPotentialDiagAdapt(3, np.zeros(3)))
    assert _user_overrides(CountingDiag(v)) == ["energy"]
    assert _host_potential_wanted(CountingDiag(v))
    assert sorted(_user_overrides(LowRankPotential(v, np.ones((3, 1))))) == ["energy", "random", "velocity", "velocity_energy"]

    class OnlyUpdate(QuadPotentialDiagAdapt):   # a per-draw hook is not a per-leapfrog method: the chain stays on the device
        def update(self, sample, grad, tune):
            return super().update(sample, grad, tune)

    assert not _host_potential_wanted(OnlyUpdate(3, np.zeros(3)))
    assert _user_overrides(OnlyUpdate(3, np.zeros(3)), ("update",)) == ["update"]

    class AdaptiveWithEnergy(QuadPotentialDiagAdapt):   # the estimators of this class live in the device chain
        def energy(self, x, velocity=None):
            return 0.0

    with pytest.raises(TypeError, match="estimators live in the device chain"):
        _host_potential_wanted(AdaptiveWithEnergy(3, np.zeros(3)))


def test_abstract_methods_and_host_arithmetic_of_the_fixed_potentials():
    """quadpotential.py:133-145 (abstract), :611-630 (Diag), :704-725 (Full), :656-677 (FullInv)."""
    base = QuadPotential()
    for call in (lambda: base.velocity(np.zeros(2)), lambda: base.energy(np.zeros(2)), base.random,
                 lambda: base.velocity_energy(np.zeros(2), np.zeros(2))):
        with pytest.raises(NotImplementedError, match="Abstract method"):
            call()
    assert base.update(np.zeros(2), np.zeros(2), True) is None and base.reset() is None

    rng = np.random.default_rng(3)
    n = 6
    x = rng.normal(size=n)
    v = rng.uniform(0.5, 2.0, size=n)
    a = rng.normal(size=(n, n))
    cov = a @ a.T + n * np.eye(n)
    for pot, C in ((QuadPotentialDiag(v), np.diag(v)), (QuadPotentialFull(cov), cov), (QuadPotentialFullInv(np.linalg.inv(cov)), cov)):
        np.testing.assert_allclose(pot.velocity(x), C @ x, rtol=1e-12)
        out = np.empty(n)
        assert pot.velocity(x, out=out) is None or pot.velocity(x, out=out) is out
        np.testing.assert_allclose(out, C @ x, rtol=1e-12)
        np.testing.assert_allclose(pot.energy(x), 0.5 * x @ C @ x, rtol=1e-12)
        np.testing.assert_allclose(pot.energy(x, velocity=out), 0.5 * x @ C @ x, rtol=1e-12)
        out2 = np.empty(n)
        np.testing.assert_allclose(pot.velocity_energy(x, out2), 0.5 * x @ C @ x, rtol=1e-12)
        np.testing.assert_allclose(out2, C @ x, rtol=1e-12)
        # random(): momentum with covariance C^-1, from the potential's own generator (test_quadpotential.py:98-135)
        pot.set_rng(np.random.default_rng(5))
        draws = np.array([pot.random() for _ in range(4000)])
        np.testing.assert_allclose(np.cov(draws.T), np.linalg.inv(C), atol=0.05 * np.abs(np.linalg.inv(C)).max() + 0.02)
    # the stream is the reference's: one rng.normal(size=n) per random()
    pot = QuadPotentialDiag(v)
    pot.set_rng(np.random.default_rng(9))
    np.testing.assert_array_equal(pot.random(), np.random.default_rng(9).normal(size=n) * (1.0 / np.sqrt(v)))


def test_a_step_with_a_user_potential_pickles_without_its_callbacks():
    """parallel.py:504-507 cloudpickles the step: the potential travels as the caller's object, the ctypes callbacks are rebuilt
    with the engine handles in the child."""
    import pickle

    from pymc_amd.step import NUTS

    spec = models.eight_schools()
    step = NUTS(model=spec, potential=CountingDiag(np.full(spec.n, 2.0)), rng=7, defer_device=True)
    back = pickle.loads(pickle.dumps(step))
    assert type(back.potential) is CountingDiag and back._host_bridge is None and back._chain_h is None
    np.testing.assert_array_equal(back.potential.v, step.potential.v)
    assert back.potential.rng.bit_generator.state == step.potential.rng.bit_generator.state


# ---- GPU ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_user_potential_is_called():
    """The reference's test, on its model: one Normal, `QuadPotentialDiag` subclass with `energy` overridden, 10 draws."""
    from pymc_amd.model_spec import ModelBuilder
    from pymc_amd.sampling import sample

    b = ModelBuilder()
    b.Normal("a", 0.0, 1.0)
    spec = b.build()
    pot = CountingDiag(np.array([1.0]))
    from pymc_amd.step import NUTS

    step = NUTS(model=spec, potential=pot, device=0)
    res = sample(10, tune=10, step=step, model=spec, chains=1, random_seed=1, progressbar=False)
    assert pot.called["energy"] == 20          # integration.py:73: once per transition, at the start state
    assert res["draws"].shape == (1, 10, 1) and np.all(np.isfinite(res["draws"]))
    assert step._scalar("single_launch") == 0   # (not the single-workgroup kernel, which knows diagonal potentials only)
    step.close()


@pytest.mark.gpu
def test_user_potential_with_library_arithmetic_samples_like_the_device_potential():
    """`CountingDiag` does, on the host, exactly what `QuadPotentialDiag` does on the device: the two chains agree in every
    integer and to rounding in the positions (the kinetic energy is a NumPy dot there, a tree reduction here)."""
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.step import NUTS

    spec = models.hier_logit(G=6, D=4, rows_per_group=30, seed=5)
    n = spec.n
    v = np.random.default_rng(2).uniform(0.02, 0.08, size=n)
    host = NUTS(model=spec, potential=CountingDiag(v), rng=3, device=0)
    dev = NUTS(model=spec, potential=QuadPotentialDiag(v), rng=3, device=0)
    for s in (host, dev):
        s.setup_chain(np.random.default_rng(11), 20, 10)
    qh = qd = RaveledVars(np.zeros(n), spec.point_map_info)
    for i in range(30):
        if i == 20:
            host.stop_tuning(); dev.stop_tuning()
        qh, sh = host.astep(qh)
        qd, sd = dev.astep(qd)
        for k in INT_KEYS:
            assert int(sh[0][k]) == int(sd[0][k]), (i, k)
        np.testing.assert_allclose(qh.data, qd.data, rtol=1e-7 if i < 8 else 2e-2, atol=1e-9 if i < 8 else 1e-3)
        np.testing.assert_allclose(sh[0]["energy"], sd[0]["energy"], rtol=1e-7 if i < 8 else 2e-2)
    assert host.rng.bit_generator.state == dev.rng.bit_generator.state
    assert host.potential.rng.bit_generator.state == dev.potential.rng.bit_generator.state
    host.close(); dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["nuts", "hmc"])
def test_wholly_user_defined_potential_matches_the_oracle_driven_by_the_same_class(method):
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.step import NUTS, HamiltonianMC

    spec = models.hier_logit(G=6, D=4, rows_per_group=30, seed=5)
    n = spec.n
    r = np.random.default_rng(4)
    d, U = r.uniform(0.02, 0.06, size=n), 0.05 * r.normal(size=(n, 3))
    f = ref_models.SpecLogpGrad(spec)
    pot_dev, pot_ref = LowRankPotential(d, U), LowRankPotential(d, U)
    if method == "nuts":
        step = NUTS(model=spec, potential=pot_dev, rng=3, device=0)
        ref = ref_sampler.RefNUTS(f, n, potential=_OracleView(pot_ref), rng=3)
    else:
        step = HamiltonianMC(model=spec, potential=pot_dev, rng=3, device=0, path_length=0.6)
        ref = ref_sampler.RefHMC(f, n, potential=_OracleView(pot_ref), rng=3, path_length=0.6)
    step.setup_chain(np.random.default_rng(11), 15, 10)
    ref.setup_chain(np.random.default_rng(11), 15, 10)
    q, qr = RaveledVars(np.zeros(n), spec.point_map_info), np.zeros(n)
    n_leapfrog = 0
    for i in range(25):
        if i == 15:
            step.stop_tuning(); ref.stop_tuning()
        q, st = step.astep(q)
        qr, sr = ref.astep(qr)
        if method == "nuts":
            for k in INT_KEYS:
                assert int(st[0][k]) == int(sr[k]), (i, k, st[0][k], sr[k])
            n_leapfrog += int(st[0]["tree_size"])
        else:
            assert int(st[0]["n_steps"]) == int(sr["n_steps"]) and bool(st[0]["accepted"]) == bool(sr["accepted"])
            n_leapfrog += int(st[0]["n_steps"])
        np.testing.assert_allclose(q.data, qr, rtol=1e-7 if i < 8 else 2e-2, atol=1e-9 if i < 8 else 1e-3)
        np.testing.assert_allclose(st[0]["energy"], sr["energy"], rtol=1e-7 if i < 8 else 2e-2)
        assert st[0]["largest_eigval"] == 2.5 and st[0]["smallest_eigval"] == 0.5   # base_hmc.py:286: potential.stats()
    assert step.rng.bit_generator.state == ref.rng.bit_generator.state
    assert pot_dev.rng.bit_generator.state == pot_ref.rng.bit_generator.state
    # the callbacks ran where the reference's integrator calls the methods (integration.py:72-73,121,134, base_hmc.py:201,239)
    c = pot_dev.called
    assert c["random"] == 25 and c["energy"] == 25 and c["update"] == 25
    assert c["velocity_energy"] == n_leapfrog
    assert c["velocity"] == 25 + n_leapfrog
    assert [t for _, t in pot_dev.seen] == [True] * 15 + [False] * 10
    np.testing.assert_array_equal(pot_dev.seen[-1][0], q.data)
    step.close()


@pytest.mark.gpu
def test_an_exception_inside_a_callback_comes_out_of_astep_and_the_step_survives():
    from pymc_amd.blocking import RaveledVars
    from pymc_amd.step import NUTS

    class Fragile(CountingDiag):
        fail_at = None

        def velocity_energy(self, x, v_out):
            self.called["velocity_energy"] += 1
            if self.called["velocity_energy"] == self.fail_at:
                raise FloatingPointError("user code gave up")
            return super().velocity_energy(x, v_out)

    spec = models.eight_schools()
    pot = Fragile(np.ones(spec.n))
    step = NUTS(model=spec, potential=pot, rng=1, device=0)
    q = RaveledVars(np.zeros(spec.n), spec.point_map_info)
    q, _ = step.astep(q)
    pot.fail_at = pot.called["velocity_energy"] + 1
    with pytest.raises(FloatingPointError, match="user code gave up"):
        step.astep(q)
    pot.fail_at = None
    for _ in range(5):   # the chain is still usable
        q, st = step.astep(q)
        assert np.all(np.isfinite(q.data))
    with pytest.raises(ValueError, match="host potential"):
        step.draw_many({info[0]: np.zeros(info[1]) for info in spec.point_map_info}, 3)
    step.close()
