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
* sampler layer (integrator, every QuadPotential, dual averaging, the NUTS tree, HMC, RNG consumption): PINNED BY
  EXECUTING THE REFERENCE.  That layer of the reference is pure NumPy/SciPy; ``tests/golden/refrun.py`` loads exactly
  those source files from /root/reference under their real module names (stand-ins only for the names they import
  from the PyTensor side) and runs ``NUTS`` / ``HamiltonianMC`` over this oracle's log-density.  The fixtures under
  ``tests/golden/`` are the outputs of those runs; ``oracle/ref_sampler.py`` reproduces every draw and statistic
  BITWISE (ten cases: five potentials, adaptation windows, HMC), checked live wherever /root/reference exists
  (``tests/test_golden.py``).  The reference's property tests are restated as well (reversibility rtol 1e-5,
  tests/step_methods/hmc/test_hmc.py:49-74; velocity/energy identities and Welford == np.var,
  tests/step_methods/hmc/test_quadpotential.py).
  What remains unpinned there: the VALUES of the start-point jitter (drawn through PyTensor RNG ops in the
  reference) -- everything downstream of a given start point is pinned.
* ESS arithmetic is third-party (arviz) in the reference and lives in
  ``pymc_amd/stats.py`` (restated from Vehtari et al. 2021): parity unpinned.
"""
