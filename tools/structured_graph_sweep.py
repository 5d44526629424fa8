"""Host-side draw of STRUCTURED priors through the lowering (the families of tests/more_models.py with drawn shapes and parameters): graph by
the reference's bodies (needs /root/reference) -> `lower_to_spec` -> the oracle's interpreter against torch autograd of the graph at one
point (1e-9).  usage: python tools/structured_graph_sweep.py <first> <last>"""
import collections
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import graph_torch as gt  # noqa: E402
import stubgraph as sg  # noqa: E402
from oracle import ref_models  # noqa: E402
from pymc_amd import lowering  # noqa: E402
from pymc_amd import model_spec as ms  # noqa: E402

pt = sg.pt


def model(case):
    rg = np.random.default_rng(123000 + case)
    pick = lambda *xs: xs[int(rg.integers(len(xs)))]      # noqa: E731
    m = sg.StubModel()
    kind = pick("grw", "ar", "zerosum", "lkj-mv", "truncated", "censored", "ordered", "icar", "mvn-cov", "mvt", "hurdle", "sde")
    s = m.HalfNormal("s", 1.0)
    if kind == "grw":
        T = int(pick(8, 30, 31, 33, 100))
        x = m.GaussianRandomWalk("x", mu=float(rg.normal() * 0.1), sigma=s, init_dist=("Normal", dict(mu=0.0, sigma=float(pick(1.0, 10.0)))), shape=(T,))
        m.Poisson("y", mu=pt.exp(x * 0.3 + 0.2), observed=rg.poisson(1.5, size=T).astype("float64"))
    elif kind == "ar":
        T, p = int(pick(12, 40, 90)), int(pick(1, 2, 3))
        const = bool(rg.random() < 0.5)
        rho = m.Normal("rho", 0.0, 0.5, shape=(p + int(const),))
        x = m.AR("x", rho=rho, sigma=s, init_dist=("Normal", dict(mu=0.0, sigma=2.0)), constant=const, shape=(T,))
        m.Normal("y", mu=x, sigma=0.4, observed=rg.normal(size=T))
    elif kind == "zerosum":
        K, N = int(pick(3, 5, 12, 32)), int(pick(20, 150))
        z = m.ZeroSumNormal("z", sigma=s, shape=(K,))
        a = m.Normal("a", 0.0, 2.0)
        m.Normal("y", mu=a + z[rg.integers(0, K, size=N)], sigma=0.7, observed=rg.normal(size=N))
    elif kind == "lkj-mv":
        n = int(pick(2, 3, 4))
        chol = m.LKJCholeskyCov("chol", n=n, eta=float(pick(1.0, 2.0, 4.0)), sd_dist=pick(("Exponential", dict(lam=1.0)), ("HalfNormal", dict(sigma=2.0))))
        mu = m.Normal("mu", 0.0, 3.0, shape=(n,))
        m.MvNormal("y", mu=mu, chol=chol, observed=rg.normal(size=(int(pick(5, 24)), n)))
    elif kind == "truncated":
        N = int(pick(10, 70))
        lam = m.Gamma("lam", 2.0, 1.0)
        lo, up = pick((None, 2.5), (0.2, None), (0.1, 3.0))
        y = rg.uniform(0.3 if lo else 0.05, 2.4 if up else 4.0, size=N)
        m.Truncated("y", ("Exponential", dict(lam=lam)), lower=lo, upper=up, observed=y)
    elif kind == "censored":
        N = int(pick(10, 60))
        mu = m.Normal("mu", 0.0, 2.0)
        raw = rg.normal(size=N)
        m.Censored("y", ("Normal", dict(mu=mu, sigma=s)), lower=-0.8, upper=0.9, observed=np.clip(raw, -0.8, 0.9))
    elif kind == "ordered":
        N, L = int(pick(30, 80)), int(pick(3, 4))
        cut = m.Normal("cut", 0.0, 2.0, shape=(L - 1,), transform="ordered")
        b = m.Normal("b", 0.0, 1.0)
        xx = rg.normal(size=N)
        m.OrderedLogistic("y", eta=b * sg.as_tensor(xx), cutpoints=cut, observed=rg.integers(0, L, size=N).astype("float64"))
    elif kind == "icar":
        r, c = pick((2, 3), (3, 4), (4, 5), (6, 7))
        n = r * c
        W = np.zeros((n, n), dtype=np.int64)
        for i in range(r):
            for j in range(c):
                k = i * c + j
                if j + 1 < c:
                    W[k, k + 1] = W[k + 1, k] = 1
                if i + 1 < r:
                    W[k, k + c] = W[k + c, k] = 1
        phi = m.ICAR("phi", W=W, sigma=s)
        m.Poisson("y", mu=pt.exp(phi + 0.5), observed=rg.poisson(2.0, size=n).astype("float64"))
    elif kind == "mvn-cov":
        k = int(pick(2, 3, 4))
        ell = m.Gamma("ell", 2.0, 2.0)
        xs = np.sort(rg.uniform(0, 3, size=k))
        K = s ** 2 * pt.exp(sg.as_tensor(-0.5 * (xs[:, None] - xs[None, :]) ** 2) / ell ** 2) + sg.as_tensor(0.1 * np.eye(k))
        m.MvNormal("y", mu=sg.as_tensor(np.zeros(k)), cov=K, observed=rg.normal(size=(int(pick(3, 12)), k)))
    elif kind == "mvt":
        n = int(pick(2, 3))
        chol = m.LKJCholeskyCov("chol", n=n, eta=2.0, sd_dist=("Exponential", dict(lam=1.0)))
        nu = m.Gamma("nu", 2.0, 0.1)
        m.MvStudentT("y", nu=nu, mu=m.Normal("mu", 0.0, 3.0, shape=(n,)), chol=chol, observed=rg.normal(size=(int(pick(6, 20)), n)))
    elif kind == "hurdle":
        N = int(pick(20, 70))
        psi = m.Beta("psi", 2.0, 2.0)
        yy = rg.gamma(2.0, 1.0, size=N) * (rg.random(N) < 0.7)
        m.HurdleGamma("y", psi=psi, alpha=m.Gamma("k", 2.0, 1.0), beta=s + 0.3, observed=yy)
    else:
        T = int(pick(15, 50))
        th = m.Normal("th", 0.0, 1.0)
        m.EulerMaruyama("x", dt=0.1, sde_fn=lambda x, th_, s_: (th_ * x - x ** 3, s_), sde_pars=(th, s), init_dist=("Normal", dict(mu=0.0, sigma=1.0)),
                        observed=np.tanh(np.sin(np.arange(T) * 0.4) * 2.0))
    return m, kind


def sweep(lo, hi):
    """-> (counts by kind and outcome, list of the cases that are not plain agreement)"""
    res, bad = collections.Counter(), []
    for case in range(lo, hi):
        try:
            m, kind = model(case)
        except Exception:   # noqa: BLE001
            res["stand-in could not build"] += 1
            bad.append((case, "build", traceback.format_exc()[-250:]))
            continue
        try:
            spec = lowering.lower_to_spec(m)
        except lowering.NotLowerable as e:
            res[f"{kind}: not lowerable"] += 1
            bad.append((case, kind, "NotLowerable", str(e)[:140]))
            continue
        except Exception:   # noqa: BLE001
            res[f"{kind}: lowering raised"] += 1
            bad.append((case, kind, "raised", traceback.format_exc()[-300:]))
            continue
        q = np.random.default_rng(case).normal(size=spec.n) * 0.3
        lp, g = gt.joint_logp_grad(m, q)
        lp2, g2 = ref_models.evaluate(spec, q)
        ok = (not np.isfinite(lp) and not np.isfinite(lp2)) or (abs(lp - lp2) <= 1e-9 * max(1.0, abs(lp)) and np.max(np.abs(g - g2)) <= 1e-9 * max(1.0, np.max(np.abs(g))))
        res[f"{kind}: ok" if ok else f"{kind}: MISMATCH"] += 1
        if not ok:
            bad.append((case, kind, "mismatch", lp, lp2, float(np.max(np.abs(g - g2)))))
        elif ms.engine_refusal(spec) is not None:
            res[f"{kind}: engine would refuse"] += 1
            bad.append((case, kind, "refusal", ms.engine_refusal(spec)))

    return res, bad


if __name__ == "__main__":
    res_, bad_ = sweep(int(sys.argv[1]), int(sys.argv[2]))
    print(dict(sorted(res_.items())))
    for b_ in bad_[:40]:
        print("  ", b_)
