"""Convergence statistics used by the benchmark metric.

The reference delegates ESS / R-hat to third-party `arviz_stats`
(pymc/stats/convergence.py:106-109; the asv harness computes
``ess / wall-seconds``, benchmarks/benchmarks/benchmarks.py:180-224).  arviz is
not installed here, so rank-normalised bulk-ESS and split-R-hat are restated
from the paper the reference links to (Vehtari, Gelman, Simpson, Carpenter,
Buerkner 2021, "Rank-normalization, folding, and localization").  PARITY
UNPINNED: the reference's tests only assert lower bounds on `az.ess`
(tests/sampler_fixtures.py:158-164).
"""

from __future__ import annotations

import numpy as np
from scipy import stats as _st


def _autocov(x):
    """FFT autocovariance along the last axis (biased, as in Stan/ArviZ)."""
    n = x.shape[-1]
    m = 1 << int(np.ceil(np.log2(2 * n)))
    xc = x - x.mean(axis=-1, keepdims=True)
    f = np.fft.rfft(xc, n=m, axis=-1)
    ac = np.fft.irfft(f * np.conj(f), n=m, axis=-1)[..., :n]
    return ac / n


def _split(x):
    """(chains, draws) -> (2*chains, draws//2)."""
    c, n = x.shape
    h = n // 2
    return np.concatenate([x[:, :h], x[:, n - h :]], axis=0)


def _z_scale(x):
    r = _st.rankdata(x.ravel(), method="average").reshape(x.shape)
    return _st.norm.ppf((r - 0.375) / (x.size + 0.25))


def _ess_raw(x):
    """Geyer initial-monotone-sequence ESS of (chains, draws)."""
    c, n = x.shape
    if n < 4:
        return float("nan")
    acov = _autocov(x)
    chain_mean = x.mean(axis=1)
    mean_var = acov[:, 0].mean() * n / (n - 1.0)
    var_plus = mean_var * (n - 1.0) / n
    if c > 1:
        var_plus += chain_mean.var(ddof=1)
    if var_plus == 0:
        return float(c * n)
    rho = np.zeros(n)
    rho[0] = 1.0
    rho[1] = 1.0 - (mean_var - acov[:, 1].mean()) / var_plus
    t = 1
    rho_even, rho_odd = 1.0, rho[1]
    while t < n - 3 and (rho_even + rho_odd) > 0:
        rho_even = 1.0 - (mean_var - acov[:, t + 1].mean()) / var_plus
        rho_odd = 1.0 - (mean_var - acov[:, t + 2].mean()) / var_plus
        if rho_even + rho_odd >= 0:
            rho[t + 1], rho[t + 2] = rho_even, rho_odd
        t += 2
    max_t = t - 2 if t >= 2 else 0
    if rho_even > 0:
        rho[max_t + 1] = rho_even
    # monotone
    tt = 1
    while tt <= max_t - 2:
        if rho[tt + 1] + rho[tt + 2] > rho[tt - 1] + rho[tt]:
            rho[tt + 1] = (rho[tt - 1] + rho[tt]) / 2.0
            rho[tt + 2] = rho[tt + 1]
        tt += 2
    tau = -1.0 + 2.0 * rho[: max_t + 1].sum() + (rho[max_t + 1] if rho_even > 0 else 0.0)
    tau = max(tau, 1.0 / np.log10(c * n))
    return c * n / tau


def ess_bulk(x):
    """Rank-normalised split bulk-ESS of one scalar parameter, x = (chains, draws)."""
    x = np.asarray(x, dtype="float64")
    if x.ndim == 1:
        x = x[None, :]
    return _ess_raw(_z_scale(_split(x)))


def rhat(x):
    """Rank-normalised split R-hat (max of bulk and folded)."""
    x = np.asarray(x, dtype="float64")
    if x.ndim == 1:
        x = x[None, :]

    def _rh(z):
        c, n = z.shape
        w = z.var(axis=1, ddof=1).mean()
        b = n * z.mean(axis=1).var(ddof=1)
        return np.sqrt(((n - 1) / n * w + b / n) / w)

    s = _split(x)
    folded = np.abs(s - np.median(s))
    return max(_rh(_z_scale(s)), _rh(_z_scale(folded)))


def min_ess_bulk(draws, max_params=None, rng=None):
    """min over parameters of bulk-ESS; draws = (chains, draws, n)."""
    draws = np.asarray(draws)
    n = draws.shape[-1]
    idx = np.arange(n)
    if max_params is not None and n > max_params:
        idx = np.sort((rng or np.random.default_rng(0)).choice(n, size=max_params, replace=False))
    vals = np.array([ess_bulk(draws[:, :, i]) for i in idx])
    return float(np.nanmin(vals)), idx[int(np.nanargmin(vals))]
