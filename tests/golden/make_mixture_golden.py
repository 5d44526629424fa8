"""Golden vectors for the mixture node: the REFERENCE's `mixture_logprob`, `Categorical.logp` and `Normal.logp` executed on fixed
inputs (tests/golden/refrun_mixture.py) -> tests/golden/mixture_reference.npz.  Run in the build container (needs /root/reference):
    python tests/golden/make_mixture_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refrun_mixture as rm  # noqa: E402


def cases():
    rng = np.random.default_rng(20240923)
    out = []
    for K, N in ((2, 40), (3, 257), (5, 64), (8, 33)):
        mu = np.sort(rng.normal(0, 3, size=K))
        sigma = rng.uniform(0.3, 2.0, size=K)
        w = rng.dirichlet(np.ones(K) * 2.0)
        comp = rng.integers(0, K, size=N)
        y = mu[comp] + sigma[comp] * rng.normal(size=N)
        y[:3] = [mu[0] - 30.0, mu[-1] + 40.0, 0.0]        # far tails: the max-shifted logsumexp matters
        c = rng.integers(0, K, size=N)
        out.append((K, N, y, w, mu, sigma, c))
    return out


if __name__ == "__main__":
    R = rm.reference()
    d = {}
    for i, (K, N, y, w, mu, sigma, c) in enumerate(cases()):
        d[f"case{i}_y"], d[f"case{i}_w"], d[f"case{i}_mu"], d[f"case{i}_sigma"], d[f"case{i}_c"] = y, w, mu, sigma, c
        d[f"case{i}_mixture_logp"] = np.asarray(R["mixture_logprob"](y, w, mu, sigma))
        d[f"case{i}_categorical_logp"] = np.asarray(R["categorical_logp"](c, w))
        d[f"case{i}_normal_logp"] = np.asarray(R["normal_logp"](y, mu[c], sigma[c]))
    # Dirichlet weights under the default simplex transform (K >= 3): value y of K - 1 elements -> w, Dirichlet.logp(w | a), the
    # transform's log-Jacobian, and forward(backward(y)) == y
    rng = np.random.default_rng(77)
    for i, K in enumerate((3, 4, 7, 16)):
        yv = rng.normal(size=K - 1) * (0.3 + i)
        a = rng.uniform(0.4, 4.0, size=K)
        w = np.asarray(R["simplex_backward"](yv))
        d[f"simplex{i}_y"], d[f"simplex{i}_a"], d[f"simplex{i}_w"] = yv, a, w
        d[f"simplex{i}_dirichlet_logp"] = np.asarray(R["dirichlet_logp"](w, a))
        d[f"simplex{i}_log_jac_det"] = np.asarray(R["simplex_log_jac_det"](yv))
        d[f"simplex{i}_forward_of_w"] = np.asarray(R["simplex_forward"](w))
    d["n_simplex"] = np.array(4)
    # Categorical.logp outside the support (-inf) and a failing parameter check
    d["cat_out_of_range"] = np.asarray(R["categorical_logp"](np.array([-1, 0, 2, 3]), np.array([0.2, 0.5, 0.3])))
    d["n_cases"] = np.array(len(cases()))
    np.savez_compressed(os.path.join(HERE, "mixture_reference.npz"), **d)
    print({k: v.shape for k, v in d.items() if k.startswith("case0")}, d["cat_out_of_range"])
