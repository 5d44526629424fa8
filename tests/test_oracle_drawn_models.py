"""The oracle's SAMPLER against the EXECUTED reference sampler on drawn models.

Every device test of the drawn-model files compares with `oracle/ref_sampler.py`; that restatement is pinned by executing the reference's
own `nuts.py` / `base_hmc.py` / `integration.py` / `quadpotential.py` / `step_sizes.py` (tests/golden/refrun.py) on the committed fixtures
and on eight schools (tests/test_golden.py).  Here the models are the drawn ones of tests/test_gpu_fuzz.py, tests/test_gpu_glm_fuzz.py and
tests/test_gpu_mixture_fuzz.py (the small ones): the reference's classes over the oracle's log-density of a drawn model, against the
oracle's sampler -- draws, every statistic and both generators bitwise.  Needs /root/reference (CPU only)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402
import refrun  # noqa: E402
import test_gpu_fuzz as tf  # noqa: E402
import test_gpu_glm_fuzz as tg  # noqa: E402
import test_gpu_mixture_fuzz as tm  # noqa: E402

from oracle import ref_models, ref_sampler  # noqa: E402

pytestmark = pytest.mark.skipif(not refrun.available(), reason="needs the reference checkout under /root/reference")


def _small(gen, cases, limit):
    out = []
    for c in cases:
        spec, desc = gen(c)
        if spec.n <= 80 and max([f.size for f in spec.factors] + [0]) <= limit:
            out.append(c)
    return out


GENERAL = _small(tf.fuzz_model, range(0, 60), 400)[:8]
GLM = [c for c in range(0, 40) if tg.glm_fuzz_model(c)[0].glm_rows.X.shape[0] <= 1000 and tg.glm_fuzz_model(c)[0].n <= 60][:5]
MIX = [c for c in range(0, 36) if tm.mixture_fuzz_model(c)[0].mixture_rows.y.size <= 300 and tm.mixture_fuzz_model(c)[0].mixture_rows.assign is None][:4]


def _compare(spec, desc, kind="nuts", **kw):
    f = ref_models.SpecLogpGrad(spec)
    point = {v.value_name: np.zeros(v.shape) for v in spec.vars}
    step, model = refrun.make_step(kind, f, point, rng=41, **kw)
    d_ref, s_ref = refrun.run_chain(step, model, np.random.default_rng(6), 14, 5)
    cls = ref_sampler.RefNUTS if kind == "nuts" else ref_sampler.RefHMC
    orc = cls(ref_models.SpecLogpGrad(spec), spec.n, rng=41, **kw)
    d_orc, s_orc = ref_sampler.run_chain(orc, np.zeros(spec.n), np.random.default_rng(6), 14, 5)
    assert np.array_equal(d_ref, d_orc), desc
    ik, fk = (mg.INT_KEYS, mg.FLT_KEYS) if kind == "nuts" else (mg.HMC_INT_KEYS, mg.HMC_FLT_KEYS)
    for k in ik + fk:
        assert [float(s[k]) for s in s_ref] == [float(s[k]) for s in s_orc], (desc, k)
    assert step.rng.bit_generator.state == orc.rng.bit_generator.state, desc
    assert step.potential.rng.bit_generator.state == orc.potential.rng.bit_generator.state, desc


@pytest.mark.parametrize("case", GENERAL)
def test_oracle_sampler_is_the_executed_reference_on_a_drawn_model_of_the_general_ir(case):
    spec, desc = tf.fuzz_model(case)
    _compare(spec, desc, max_treedepth=7)


@pytest.mark.parametrize("case", GLM)
def test_oracle_sampler_is_the_executed_reference_on_a_drawn_model_around_the_glm_node(case):
    spec, desc = tg.glm_fuzz_model(case)
    _compare(spec, desc, max_treedepth=7)


@pytest.mark.parametrize("case", MIX)
def test_oracle_sampler_is_the_executed_reference_on_a_drawn_model_around_the_mixture_node(case):
    spec, desc = tm.mixture_fuzz_model(case)
    _compare(spec, desc, max_treedepth=6)


@pytest.mark.parametrize("case", GENERAL[:3])
def test_oracle_hmc_is_the_executed_reference_on_a_drawn_model(case):
    spec, desc = tf.fuzz_model(case)
    _compare(spec, desc, kind="hmc", path_length=1.0, max_steps=9)


# ---- the oracle's Gibbs sweep against the executed `CategoricalGibbsMetropolis` on drawn mixtures ---------------------------------------
@pytest.mark.parametrize("case", list(range(12)))
@pytest.mark.parametrize("proposal", ["uniform"])
def test_oracle_gibbs_sweeps_are_the_executed_reference_on_a_drawn_mixture(case, proposal):
    """`metropolis.py:761-786` (`astep_unif`, the default proposal) executed over the full-model log-density of a DRAWN mixture -- K = 2 .. 9
    components, unequal weights, a scale per component, 40 .. 400 rows -- against `oracle/ref_gibbs.py`: assignments after every sweep
    and the generator's state (the committed fixtures hold one shape: K = 3, N = 240, equal weights, one scale).

    `proposal="proportional"` is NOT drawn, and the reason is a finding: `metropolis_proportional` (:805-826) divides by
    `1 - prob_curr` and by `1 - probs[proposed]`, and rejects WITHOUT drawing its uniform when the ratio is not finite.  On separated
    mixtures those differences round to exactly 0 or to 1e-16 depending on the last bits of the softmax's inputs -- the full-model
    log-densities in the reference, the per-row terms in the restatement (equal in exact arithmetic) -- so one side consumes a uniform
    the other does not and the streams part (K = 3, N = 400: 23 assignments of one sweep); three of eight drawn shapes made the
    reference's own `rng.choice` raise ("Probabilities contain NaN" / "do not sum to 1").  The committed K = 3 / N = 240 fixture does not
    meet the event, which is all its bitwise agreement says."""
    import types

    import make_gibbs_golden as mgg
    from oracle import ref_gibbs

    rg = np.random.default_rng(4200 + case)
    K = int(rg.integers(2, 10))
    N = int(rg.choice([40, 97, 240, 400]))
    w = rg.dirichlet(np.ones(K) * 2.0)
    sigma = rg.uniform(0.5, 1.6, size=K)
    mu_true = np.linspace(-3.0, 3.0, K)
    c_true = rg.choice(K, size=N, p=w)
    y = mu_true[c_true] + sigma[c_true] * rg.normal(size=N)
    link = types.SimpleNamespace(y=y, K=K, log_w=np.log(w), sigma=sigma)
    spec = types.SimpleNamespace(mixture=link)
    n_sweeps = 3
    mus = mu_true[None, :] + 0.4 * rg.normal(size=(n_sweeps, K))
    c0 = rg.integers(0, K, size=N)
    seed = 900 + case
    cs, state = mgg.reference_sweeps(spec, seed, n_sweeps, mus, c0, leave_half_cached=True, proposal=proposal)
    rng = np.random.default_rng(seed)
    rng.integers(2**30)
    g = ref_gibbs.RefCategoricalGibbs(link.y, link.log_w, link.sigma, rng)
    c = c0.copy()
    for s_ in range(n_sweeps):
        c, _ = (g.sweep if proposal == "uniform" else g.sweep_prop)(c, mus[s_])
        assert np.array_equal(c, cs[s_]), (case, proposal, s_)
    assert rng.bit_generator.state == state, (case, proposal)


# ---- the oracle's ADVI step against the executed reference variational code on drawn GLM shapes -----------------------------------------
@pytest.mark.parametrize("case", list(range(8)))
def test_oracle_advi_steps_are_the_executed_reference_on_a_drawn_glm(case):
    """`tests/golden/refrun_advi.py` executes the reference's `FullRankGroup` / `KL.apply` / normalised terms / `adagrad_window` / `logp`
    bodies eagerly; the committed fixture is ONE shape (400 x 6, batches of 16).  Here the shape is drawn -- rows, covariates, batch size,
    noise and prior scales, learning rate, family -- and the oracle (`oracle/ref_advi.py`), fed the same rows and standard-normal draws,
    must give the same loss, gradients, parameters and rings over 13 steps (the window of 10 wraps)."""
    import refrun_advi as ra
    from oracle import ref_advi

    if not ra.available():
        pytest.skip("needs /root/reference")
    rg = np.random.default_rng(8100 + case)
    N, P, B = int(rg.choice([60, 400, 1500])), int(rg.choice([1, 3, 8, 17])), int(rg.choice([4, 16, 50]))
    fam = ("normal", "bernoulli")[case % 2]
    sigma, prior_sd, lr = float(rg.uniform(0.4, 1.5)), float(rg.uniform(0.8, 3.0)), float(rg.choice([0.01, 0.05, 0.2]))
    X = rg.normal(size=(N, P))
    if rg.random() < 0.5:
        X[:, 0] = 1.0
    beta = rg.normal(size=P)
    y = X @ beta + rg.normal(size=N) * sigma if fam == "normal" else (rg.uniform(size=N) < 1 / (1 + np.exp(-X @ beta))).astype("float64")
    steps = 13
    idx, z0 = rg.integers(0, N, size=(steps, B)), rg.normal(size=(steps, P))
    ref = ra.Stepper(X, y, fam, sigma=sigma, prior_sd=prior_sd)
    glm = ref_advi.GLM(X, y, fam, sigma, prior_sd)
    st = ref_advi.FullRankState(P)
    for s_ in range(steps):
        loss, gm, gl = ref.step(idx[s_], z0[s_], learning_rate=lr)
        l, om, ol = ref_advi.advi_step(glm, st, idx[s_], z0[s_], learning_rate=lr)
        # (the LOSS to 1e-9 only: the reference's Bernoulli density is `switch(y, log(p), log1p(-p))` of p = sigmoid(eta) (discrete.py:351-377), which
        # loses digits of log1p(-p) once a row of the batch saturates (|eta| ~ 17: 1 - p = 4e-8 known to 1e-16), the oracle's is the softplus
        # form; measured 1e-11 relative at two of thirteen steps of the 1 500 x 17 shapes, 1e-15 elsewhere.  Gradients and parameters -- what
        # the optimisation consumes -- agree to 1e-15 at every step and are held to 1e-12)
        assert abs(l - loss) <= 1e-9 * max(1.0, abs(loss)), (case, s_)
        (am, im), (aL, iL) = ref.ring()
        for got, want in ((om, gm), (ol, gl), (st.mu, ref.mu), (st.L_tril, ref.L_tril), (st.acc_mu, am), (st.acc_L, aL)):
            assert np.max(np.abs(np.asarray(got) - np.asarray(want))) <= 1e-12 * max(1.0, np.max(np.abs(want))), (case, s_)
        assert st.i == int(im)


# ---- every adapting potential on drawn models ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", GENERAL[:4])
@pytest.mark.parametrize("potential", ["adapt_diag", "full", "fullinv", "full_adapt", "diag_adapt_exp"])
def test_oracle_potentials_are_the_executed_reference_on_a_drawn_model(case, potential):
    """The five `QuadPotential` classes the fixtures pin on eight schools / a standard normal (quadpotential.py: DiagAdapt, Full, FullInv,
    FullAdapt, DiagAdaptExp with gradients) under NUTS on a drawn model: 16 tuning transitions (FullAdapt refreshes its covariance and
    factor at every one past the first window) + 4 draws, bitwise."""
    import warnings

    spec, desc = tf.fuzz_model(case)
    ref = refrun.load()
    f = ref_models.SpecLogpGrad(spec)
    rngs, seeds = ref_sampler.spawn_chain_rngs(77 + case, 1)
    mk_ref, mk_orc = mg._potentials(dict(potential=potential), spec.n, seeds[0])
    point = {v.value_name: np.zeros(v.shape) for v in spec.vars}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # ("QuadPotentialFullAdapt is an experimental feature")
        step, model = refrun.make_step("nuts", f, point, potential=mk_ref(ref.quadpotential), rng=seeds[0], max_treedepth=7)
        d_ref, s_ref = refrun.run_chain(step, model, rngs[0], 16, 4)
        rngs2, seeds2 = ref_sampler.spawn_chain_rngs(77 + case, 1)
        orc = ref_sampler.RefNUTS(ref_models.SpecLogpGrad(spec), spec.n, potential=mk_orc(), rng=seeds2[0], max_treedepth=7)
        d_orc, s_orc = ref_sampler.run_chain(orc, np.zeros(spec.n), rngs2[0], 16, 4)
    assert np.array_equal(d_ref, d_orc), (desc, potential)
    for k in mg.INT_KEYS + mg.FLT_KEYS:
        assert [float(s[k]) for s in s_ref] == [float(s[k]) for s in s_orc], (desc, potential, k)


@pytest.mark.parametrize("case", GENERAL[4:7])
@pytest.mark.parametrize(
    "kind,kw",
    [
        ("nuts", dict(adapt_step_size=False, step_scale=0.1, max_treedepth=6)),
        ("nuts", dict(Emax=5.0, early_max_treedepth=3, max_treedepth=5)),       # divergences and depth-limited trees
        ("nuts", dict(gamma=0.1, k=0.6, t0=5, target_accept=0.9, max_treedepth=6)),
        ("hmc", dict(path_length=3.0, adapt_step_size=False, step_scale=0.2, max_steps=12)),
    ],
)
def test_oracle_step_options_mean_what_the_references_mean_on_a_drawn_model(case, kind, kw):
    """The constructor options of `BaseHMC` / `NUTS` / `HamiltonianMC` (base_hmc.py:82-103, nuts.py:132-147, hmc.py:70-77) on drawn models
    (tests/test_golden.py holds them on eight schools)."""
    spec, desc = tf.fuzz_model(case)
    _compare(spec, desc, kind=kind, **kw)
