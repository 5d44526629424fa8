#!/usr/bin/env python
"""Which coordinate of C2-L does not mix, and why?  (VERDICT r01, weak #2.)

Runs `configs[1]` at the benchmarked size on the device, 4 chains x (1000 tune + 1000 draws), and prints one JSON
object: bulk-ESS / R-hat of every parameter, the worst ones by name, and the diagnostics of the non-centred ridge
(mu_d against the group mean of z_{.,d}): the data determine beta_{g,d} = mu_d + sigma_d z_{g,d} sharply (4000 rows per
group), so (mu_d, mean_g z_{g,d}) is a long thin ridge that a DIAGONAL metric cannot rescale -- the well-identified
combination mixes, the two pieces do not.  Run on the GPU box:  python tools/ess_study.py [rows_per_group]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pymc_amd import models, stats
    from pymc_amd.sampling import sample

    rpg = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    G, D = 1248, 8
    spec = models.hier_logit(G=G, D=D, rows_per_group=rpg)
    t0 = time.time()
    res = sample(draws=1000, tune=1000, chains=4, model=spec, init="jitter+adapt_diag", random_seed=20160911, device=0)
    wall = time.time() - t0
    d = res["draws"]                                  # (4, 1000, n)
    ess, rh = stats.ess_bulk_many(d), stats.rhat_many(d)
    names = []
    for v in spec.vars:
        names += [f"{v.value_name}[{k}]" for k in range(v.size)]
    order = np.argsort(ess)
    mu, ls, z = d[:, :, :D], d[:, :, D : 2 * D], d[:, :, 2 * D :].reshape(4, -1, G, D)
    zbar = z.mean(axis=2)                             # (4, draws, D)
    beta_bar = mu + np.exp(ls) * zbar                 # the combination the likelihood pins down
    ridge = []
    for k in range(D):
        c = np.corrcoef(mu[:, :, k].ravel(), zbar[:, :, k].ravel())[0, 1]
        ridge.append({
            "d": k, "corr_mu_zbar": float(c),
            "ess_mu": float(ess[k]), "ess_sigma_log": float(ess[D + k]),
            "ess_zbar": float(stats.ess_bulk(zbar[:, :, k])), "ess_beta_bar": float(stats.ess_bulk(beta_bar[:, :, k])),
            "sd_mu": float(mu[:, :, k].std()), "sd_beta_bar": float(beta_bar[:, :, k].std()),
            "rhat_mu": float(rh[k]), "rhat_beta_bar": float(stats.rhat(beta_bar[:, :, k])),
        })
    tree = np.array([[s["tree_size"] for s in chain] for chain in res["stats"]])
    out = {
        "workload": f"hier-logit G={G} D={D} rows_per_group={rpg} n={spec.n}, 4 chains x (1000 + 1000), jitter+adapt_diag",
        "wall_s": wall, "min_ess": float(ess.min()), "median_ess": float(np.median(ess)), "ess_5pct": float(np.percentile(ess, 5)),
        "rhat_max": float(rh.max()), "n_rhat_gt_1.01": int((rh > 1.01).sum()), "n_rhat_gt_1.1": int((rh > 1.1).sum()),
        "worst": [{"param": names[j], "ess": float(ess[j]), "rhat": float(rh[j])} for j in order[:12]],
        "ess_z_min": float(ess[2 * D :].min()), "ess_z_median": float(np.median(ess[2 * D :])),
        "ridge": ridge,
        "mean_tree_size": float(tree.mean()), "tree_size_per_chain": [float(x) for x in tree.mean(axis=1)],
        "step_size_bar": [float(chain[-1]["step_size_bar"]) for chain in res["stats"]],
        "divergences": [int(chain[-1]["divergences"]) for chain in res["stats"]],
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
