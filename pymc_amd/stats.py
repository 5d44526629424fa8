"""Convergence statistics used by the benchmark metric.

The reference delegates ESS / R-hat to third-party `arviz_stats`
(pymc/stats/convergence.py:106-109; the asv harness computes
``ess / wall-seconds``, benchmarks/benchmarks/benchmarks.py:180-224).  arviz is
not installed here, so rank-normalised bulk-ESS and split-R-hat are restated
from the paper the reference links to (Vehtari, Gelman, Simpson, Carpenter,
Buerkner 2021, "Rank-normalization, folding, and localization").  The reference's
own tests only assert lower bounds on `az.ess` (tests/sampler_fixtures.py:158-164),
so there is no golden vector to take from it; the estimators are pinned against an
independent restatement (direct autocovariance, 1-based lags as the `posterior`
package writes the algorithm, explicit ranks) on fixed arrays committed under
tests/golden/stats_kat.npz (generator tests/golden/make_stats_golden.py), and against
the closed-form ESS of iid and AR(1) processes (tests/test_stats.py).
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
    tau = -1.0 + 2.0 * rho[: max_t + 1].sum() + rho[max_t + 1]
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


# ---------------------------------------------------------------------------
# the same estimators for MANY parameters at once (the benchmark's min-over-parameters ESS runs over all n = 10 000
# coordinates; one Python-level pass per coordinate took longer than the sampling it summarises)
# ---------------------------------------------------------------------------


def _z_scale_many(x):
    """x (..., P): rank-normalise each column P over ALL its leading entries."""
    lead = x.shape[:-1]
    flat = x.reshape(-1, x.shape[-1])
    r = _st.rankdata(flat, method="average", axis=0)
    return _st.norm.ppf((r - 0.375) / (flat.shape[0] + 0.25)).reshape(lead + (x.shape[-1],))


def _split_many(x):
    """(chains, draws, P) -> (2 chains, draws // 2, P)."""
    c, n, p = x.shape
    h = n // 2
    return np.concatenate([x[:, :h], x[:, n - h :]], axis=0)


def _ess_raw_many(x):
    """`_ess_raw` for x = (chains, draws, P); returns (P,).  Same arithmetic, vectorised over P."""
    c, n, P = x.shape
    if n < 4:
        return np.full(P, np.nan)
    m = 1 << int(np.ceil(np.log2(2 * n)))
    xc = x - x.mean(axis=1, keepdims=True)
    f = np.fft.rfft(xc, n=m, axis=1)
    acov = np.fft.irfft(f * np.conj(f), n=m, axis=1)[:, :n] / n          # (c, n, P)
    chain_mean = x.mean(axis=1)                                           # (c, P)
    mean_var = acov[:, 0].mean(axis=0) * n / (n - 1.0)                    # (P,)
    var_plus = mean_var * (n - 1.0) / n
    if c > 1:
        var_plus = var_plus + chain_mean.var(axis=0, ddof=1)
    out = np.empty(P)
    zero = var_plus == 0
    vp = np.where(zero, 1.0, var_plus)
    rho_all = 1.0 - (mean_var[None, :] - acov.mean(axis=0)) / vp[None, :]  # (n, P)
    rho_all[0] = 1.0
    for j in range(P):   # the Geyer truncation is a data-dependent scan; it touches a handful of lags per parameter
        if zero[j]:
            out[j] = float(c * n)
            continue
        r = rho_all[:, j]
        rho = np.zeros(n)
        rho[0] = 1.0
        rho[1] = r[1]
        t = 1
        rho_even, rho_odd = 1.0, r[1]
        while t < n - 3 and (rho_even + rho_odd) > 0:
            rho_even, rho_odd = r[t + 1], r[t + 2]
            if rho_even + rho_odd >= 0:
                rho[t + 1], rho[t + 2] = rho_even, rho_odd
            t += 2
        max_t = t - 2 if t >= 2 else 0
        if rho_even > 0:
            rho[max_t + 1] = rho_even
        tt = 1
        while tt <= max_t - 2:
            if rho[tt + 1] + rho[tt + 2] > rho[tt - 1] + rho[tt]:
                rho[tt + 1] = (rho[tt - 1] + rho[tt]) / 2.0
                rho[tt + 2] = rho[tt + 1]
            tt += 2
        tau = -1.0 + 2.0 * rho[: max_t + 1].sum() + rho[max_t + 1]
        tau = max(tau, 1.0 / np.log10(c * n))
        out[j] = c * n / tau
    return out


def ess_bulk_many(draws, block=2048):
    """Rank-normalised split bulk-ESS of every parameter; draws = (chains, draws, P) -> (P,)."""
    draws = np.asarray(draws, dtype="float64")
    if draws.ndim == 2:
        draws = draws[None]
    P = draws.shape[-1]
    out = np.empty(P)
    for s in range(0, P, block):
        out[s : s + block] = _ess_raw_many(_z_scale_many(_split_many(draws[:, :, s : s + block])))
    return out


def rhat_many(draws, block=2048):
    """Rank-normalised split R-hat (max of bulk and folded) of every parameter; draws = (chains, draws, P) -> (P,)."""
    draws = np.asarray(draws, dtype="float64")
    if draws.ndim == 2:
        draws = draws[None]
    P = draws.shape[-1]
    out = np.empty(P)

    def _rh(z):
        n = z.shape[1]
        w = z.var(axis=1, ddof=1).mean(axis=0)
        b = n * z.mean(axis=1).var(axis=0, ddof=1)
        return np.sqrt(((n - 1) / n * w + b / n) / w)

    for s in range(0, P, block):
        sp = _split_many(draws[:, :, s : s + block])
        folded = np.abs(sp - np.median(sp.reshape(-1, sp.shape[-1]), axis=0))
        out[s : s + block] = np.maximum(_rh(_z_scale_many(sp)), _rh(_z_scale_many(folded)))
    return out
