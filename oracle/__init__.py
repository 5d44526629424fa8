"""CPU oracle for the NUTS/HMC hot path of pymc-devs/pymc.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from here.  ``pymc_amd`` never imports it; the
product path runs on hand-written HIP kernels and fails loudly without them.

What it is
----------
A plain NumPy/SciPy restatement of the reference's sampler semantics (the
reference itself is pure Python on top of PyTensor; PyTensor, arviz and
Python>=3.12 are absent from the build image, so ``import pymc`` is impossible
here -- SURVEY.md section 8c).  Every function cites the reference file:line it
follows (paths relative to the reference checkout).

Parity status
-------------
* logp / gradient layer: PINNED by the reference's own known answers --
  joint logp ``-12.691227342634292`` (pymc/pytensorf.py:514-546), the
  ``ValueGradFunction`` literals (tests/model/test_core.py:386-421,457-465),
  SciPy logpdf/logpmf agreement to 6 decimals (pymc/testing.py:311-417) and
  torch float64 autograd for every hand-written gradient
  (tests/test_oracle_*.py).
* integrator / potentials / dual averaging: PINNED by the reference's
  property tests restated in tests/ (reversibility rtol 1e-5,
  tests/step_methods/hmc/test_hmc.py:49-74; velocity/energy identities and
  Welford == np.var, tests/step_methods/hmc/test_quadpotential.py).
* NUTS draw sequences: PARITY UNPINNED.  The reference holds no golden draw
  vectors (all of its NUTS assertions are statistical) and
  cannot be executed here to generate any.  The tree restatement is pinned only
  by the reference's statistical fixtures (tests/sampler_fixtures.py) and by
  line-by-line review against pymc/step_methods/hmc/nuts.py:204-489.
* ESS arithmetic is third-party (arviz) in the reference and lives in
  ``pymc_amd/stats.py`` (restated from Vehtari et al. 2021): parity unpinned.
"""
