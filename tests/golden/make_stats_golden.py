"""Known-answer vectors for `pymc_amd/stats.py` (rank-normalised split bulk-ESS, split R-hat).

The reference delegates both to third-party arviz (pymc/stats/convergence.py:106-109; absent here), so the estimator in
`pymc_amd/stats.py` (FFT autocovariance, 0-based scan after ArviZ's `_ess`) is pinned against an INDEPENDENT
restatement kept in this file: the algorithm as the `posterior` R package / Stan reference manual write it down
(`ess_rfun`: 1-based lags, Geyer's initial positive + monotone sequence), with a direct O(n^2) autocovariance and
ranks by explicit sorting.  The two share no code.  Fixed input arrays and this restatement's answers are committed
in `stats_kat.npz`; `tests/test_stats.py` checks `pymc_amd.stats` against them (and against closed forms for iid /
AR(1) processes).

    python tests/golden/make_stats_golden.py
"""

import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _phi_inv(p):
    """Acklam's rational approximation of the normal quantile + one Halley step (independent of scipy's ndtri)."""
    a = [-3.969683028665376e01, 2.209460984245205e02, -2.759285104469687e02, 1.383577518672690e02, -3.066479806614716e01, 2.506628277459239e00]
    b = [-5.447609879822406e01, 1.615858368580409e02, -1.556989798598866e02, 6.680131188771972e01, -1.328068155288572e01]
    c = [-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e00, -2.549732539343734e00, 4.374664141464968e00, 2.938163982698783e00]
    d = [7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e00, 3.754408661907416e00]
    if p < 0.02425:
        q = math.sqrt(-2 * math.log(p))
        x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1)
    elif p > 1 - 0.02425:
        q = math.sqrt(-2 * math.log(1 - p))
        x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1)
    else:
        q = p - 0.5
        r = q * q
        x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1)
    e = 0.5 * math.erfc(-x / math.sqrt(2)) - p
    u = e * math.sqrt(2 * math.pi) * math.exp(x * x / 2)
    return x - u / (1 + x * u / 2)


def _avg_ranks(v):
    order = sorted(range(len(v)), key=lambda i: v[i])
    ranks = [0.0] * len(v)
    i = 0
    while i < len(v):
        j = i
        while j + 1 < len(v) and v[order[j + 1]] == v[order[i]]:
            j += 1
        for k in range(i, j + 1):
            ranks[order[k]] = (i + j) / 2.0 + 1.0
        i = j + 1
    return ranks


def z_scale(chains):
    flat = [x for ch in chains for x in ch]
    r = _avg_ranks(flat)
    S = len(flat)
    z = [_phi_inv((ri - 3.0 / 8.0) / (S - 2.0 * 3.0 / 8.0 + 1.0)) for ri in r]
    n = len(chains[0])
    return [z[k * n : (k + 1) * n] for k in range(len(chains))]


def split(chains):
    out = []
    n = len(chains[0])
    h = n // 2
    for ch in chains:
        out.append(list(ch[:h]))
    for ch in chains:
        out.append(list(ch[n - h :]))
    return out


def _acov(ch):
    n = len(ch)
    m = sum(ch) / n
    return [sum((ch[i] - m) * (ch[i + t] - m) for i in range(n - t)) / n for t in range(n)]


def ess_rfun(chains):
    """`posterior:::ess_rfun` transcribed with its 1-based indices kept (acov[t] below is R's acov[t, ])."""
    nchain, niter = len(chains), len(chains[0])
    ac = [_acov(ch) for ch in chains]
    acov = [None] + [[ac[k][t] for k in range(nchain)] for t in range(niter)]   # acov[1] = lag 0
    mean = lambda v: sum(v) / len(v)
    chain_mean = [sum(ch) / niter for ch in chains]
    mean_var = mean(acov[1]) * niter / (niter - 1)
    var_plus = mean_var * (niter - 1) / niter
    if nchain > 1:
        cm = mean(chain_mean)
        var_plus += sum((x - cm) ** 2 for x in chain_mean) / (nchain - 1)
    rho = [0.0] * (niter + 2)
    t = 0
    even = 1.0
    rho[t + 1] = even
    odd = 1 - (mean_var - mean(acov[t + 2])) / var_plus
    rho[t + 2] = odd
    # truncation bound: `posterior` stops at t < niter - 5; ArviZ (`t < n_draw - 3` with its t one ahead), which is what the
    # reference calls (pymc/stats/convergence.py:106-109), looks one pair of lags further -- the KAT follows ArviZ.  The two
    # differ only when the autocorrelation never turns negative within the chain (e.g. chains that disagree).
    while t < niter - 4 and not math.isnan(even + odd) and even + odd > 0:
        t += 2
        even = 1 - (mean_var - mean(acov[t + 1])) / var_plus
        odd = 1 - (mean_var - mean(acov[t + 2])) / var_plus
        if even + odd >= 0:
            rho[t + 1] = even
            rho[t + 2] = odd
    max_t = t
    if even > 0:
        rho[max_t + 1] = even
    t = 0
    while t <= max_t - 4:
        t += 2
        if rho[t + 1] + rho[t + 2] > rho[t - 1] + rho[t]:
            rho[t + 1] = (rho[t - 1] + rho[t]) / 2
            rho[t + 2] = rho[t + 1]
    ess = nchain * niter
    tau = -1 + 2 * sum(rho[1 : max_t + 1]) + rho[max_t + 1]
    tau = max(tau, 1 / math.log10(ess))
    return ess / tau


def ess_bulk(chains):
    return ess_rfun(z_scale(split(chains)))


def _rhat_basic(chains):
    m, n = len(chains), len(chains[0])
    means = [sum(c) / n for c in chains]
    vars_ = [sum((x - mu) ** 2 for x in c) / (n - 1) for c, mu in zip(chains, means)]
    gm = sum(means) / m
    B = n * sum((mu - gm) ** 2 for mu in means) / (m - 1)
    W = sum(vars_) / m
    return math.sqrt(((n - 1) / n * W + B / n) / W)


def rhat(chains):
    s = split(chains)
    flat = sorted(x for c in s for x in c)
    k = len(flat)
    med = flat[k // 2] if k % 2 else 0.5 * (flat[k // 2 - 1] + flat[k // 2])
    folded = [[abs(x - med) for x in c] for c in s]
    return max(_rhat_basic(z_scale(s)), _rhat_basic(z_scale(folded)))


def main():
    rng = np.random.default_rng(20160911)
    cases = {}
    iid = rng.normal(size=(4, 120))
    ar = rng.normal(size=(4, 160))
    for k in range(1, ar.shape[1]):
        ar[:, k] = 0.7 * ar[:, k - 1] + math.sqrt(1 - 0.49) * ar[:, k]
    anti = rng.normal(size=(2, 100))
    for k in range(1, anti.shape[1]):
        anti[:, k] = -0.5 * anti[:, k - 1] + anti[:, k]
    shifted = rng.normal(size=(3, 90)) + np.array([[0.0], [0.6], [-0.3]])     # chains that disagree: R-hat > 1
    drift = np.cumsum(rng.normal(size=(1, 200)), axis=1)                       # one chain, random walk: tiny ESS
    ties = np.round(rng.normal(size=(2, 80)), 1)                               # tied values: average ranks
    for name, x in [("iid", iid), ("ar1", ar), ("anti", anti), ("shifted", shifted), ("drift", drift), ("ties", ties)]:
        ch = [list(map(float, row)) for row in x]
        cases[name + "_x"] = x
        cases[name + "_ess_bulk"] = ess_bulk(ch)
        cases[name + "_rhat"] = rhat(ch) if x.shape[0] > 1 else rhat(ch)
        print(name, cases[name + "_ess_bulk"], cases[name + "_rhat"])
    np.savez_compressed(os.path.join(HERE, "stats_kat.npz"), **cases)


if __name__ == "__main__":
    main()
