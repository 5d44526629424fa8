"""Writes tests/golden/advi_reference_steps.npz: 14 consecutive optimisation steps of full-rank minibatch ADVI on a small GLM
(Normal and Bernoulli likelihood), every arithmetic step taken by THE REFERENCE's own code executed eagerly
(tests/golden/refrun_advi.py).  Per step: loss, d loss / d mu, d loss / d L_tril, the parameters and both `adagrad_window` rings
after the update.

    python tests/golden/make_advi_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def run():
    import refrun_advi as ra

    rng = np.random.default_rng(20160911)
    N, P, B, steps, lr = 400, 6, 16, 14, 0.05
    X = rng.normal(size=(N, P))
    X[:, 0] = 1.0
    beta = rng.normal(size=P)
    out = {"X": X, "y_normal": X @ beta + rng.normal(size=N) * 0.7, "y_bernoulli": (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ beta))).astype("float64"),
           "sigma": np.array(0.7), "prior_sd": np.array(2.0), "learning_rate": np.array(lr),
           "idx": rng.integers(0, N, size=(steps, B)), "z0": rng.normal(size=(steps, P))}
    for fam in ("normal", "bernoulli"):
        st = ra.Stepper(X, out[f"y_{fam}"], fam, sigma=0.7, prior_sd=2.0)
        rec = {k: [] for k in ("loss", "grad_mu", "grad_L", "mu", "L", "ring_mu", "ring_L", "ring_i")}
        for s in range(steps):
            loss, gm, gl = st.step(out["idx"][s], out["z0"][s], learning_rate=lr)
            (am, im), (aL, iL) = st.ring()
            assert im == iL
            for k, v in zip(rec, (loss, gm, gl, st.mu.copy(), st.L_tril.copy(), am.copy(), aL.copy(), im)):
                rec[k].append(v)
        out.update({f"{fam}_{k}": np.array(v) for k, v in rec.items()})
    return out


if __name__ == "__main__":
    path = os.path.join(HERE, "advi_reference_steps.npz")
    np.savez_compressed(path, **run())
    print(path, os.path.getsize(path), "bytes")
