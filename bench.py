#!/usr/bin/env python
"""Benchmark of the NUTS hot path on MI355X (contract: see the round prompt, section 4).

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2-L hier-logit-10k"):
hierarchical logistic regression, n = 10 000 parameters (G=1248 groups x D=8),
N = 4 992 000 observation rows (X fp64 = 319.5 MB > 256 MiB LLC: HBM regime),
one NUTS chain per GPU, reference sampler defaults (target_accept 0.8,
max_treedepth 10 / 8 early, jitter+adapt_diag).

A "step" is one NUTS draw.  `--warmup W` tuning draws run untimed, then EXACTLY
`--steps K` post-warmup draws are timed between barrier + device synchronisation.
value = aggregate effective samples/s = sum over chains of min-over-parameters
bulk-ESS of the K timed draws / max-over-ranks wall time.  leapfrog steps/s
(= sum tree_size / time) is reported next to it.

roofline: dominant kernel = k_rows (csrc/rows_kernel.h); algorithmic bytes per
launch = 69 B/row x N (SURVEY.md 8d B_model), divided by the kernel's average
duration measured with HIP events on the library stream during the timed region
(1 launch in 8 is bracketed by events; launches that drain after the tree
terminated are included, exactly as in the rocprofv3 summary under profiles/).
`traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes
(profiles/traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
MI355X_MICROARCH.md), not collected inside this run.

cpu_baseline (rank 0, N=1 only): the oracle's reference-order leapfrog with the
single-threaded C restatement of the logp+grad (stand-in for PyTensor's C linker,
which cannot run here), timed on a bounded number of leapfrogs of the same
workload; kind = "port", cores = 1.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--rows-per-group", type=int, default=4000, help="4000 = C2-L (HBM regime), 80 = C2-S (cache resident)")
    ap.add_argument("--groups", type=int, default=1248)
    ap.add_argument("--seed", type=int, default=20160911)
    ap.add_argument("--cpu-leapfrogs", type=int, default=150, help="bounded CPU-baseline sample, about 18 s of one host core (0 disables)")
    ap.add_argument("--ess-params", type=int, default=1500, help="parameters sampled for the min-ESS (all of mu/sigma + random z)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for tests)")
    ap.add_argument("--share-gpu", action="store_true", help="TEST ONLY: every rank uses GPU 0 (exercises the N > 1 logic on a 1-GPU box)")
    return ap.parse_args()


def cpu_baseline(spec, q, step_size, inv_mass, n_leap, ess_per_leapfrog):
    """Reference-order leapfrogs (oracle/ref_sampler.Leapfrog) on one host core."""
    from oracle import c_logit, ref_sampler

    try:
        import threadpoolctl

        ctx = threadpoolctl.threadpool_limits(1)
    except Exception:  # pragma: no cover
        ctx = None
    f = c_logit.CHierLogit(spec, so_path=c_logit.build_native())   # compiled for the box it is timed on
    pot = ref_sampler.DiagPotential(inv_mass)
    integ = ref_sampler.Leapfrog(pot, f)
    rng = np.random.default_rng(0)
    s = integ.compute_state(np.array(q), pot.random() if False else rng.normal(size=spec.n) / np.sqrt(inv_mass))
    t0 = time.perf_counter()
    for _ in range(n_leap):
        s = integ.step(step_size, s)
    dt = time.perf_counter() - t0
    lps = n_leap / dt
    return {
        "value": lps * ess_per_leapfrog,
        "unit": "ESS/s",
        "leapfrog_steps_per_sec": lps,
        "cores": 1,
        "kind": "port",
        "host_cores_available": os.cpu_count(),
        "sample": f"{n_leap} reference-order leapfrog steps (oracle integrator + single-threaded gcc -O3 -march=native fused logp/grad, compiled on this box) "
        f"of the same workload in {dt:.1f} s; ESS/s = leapfrog/s x the GPU run's measured ESS per leapfrog ({ess_per_leapfrog:.4g})",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    if args.share_gpu:
        local = 0
        args.backend = "gloo"   # RCCL refuses two ranks on one device ("Duplicate GPU detected")
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    comm_dev = "cuda" if args.backend == "nccl" else "cpu"

    from pymc_amd import models
    from pymc_amd.sampling import init_nuts
    from pymc_amd.step import get_random_generator
    from pymc_amd.stats import ess_bulk

    spec = models.hier_logit(G=args.groups, D=8, rows_per_group=args.rows_per_group, seed=20160911)
    N = spec.logit_rows.X.shape[0]
    chains = world
    rngs = get_random_generator(args.seed).spawn(chains)  # mcmc.py:907-908
    seed_list = [int(r.integers(2**30)) for r in rngs]
    points, step = init_nuts(spec, init="jitter+adapt_diag", chains=chains, random_seed_list=seed_list, device=local)
    alg_bytes = step._logp_dlogp_func.algorithmic_bytes

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    W, K = args.warmup, args.steps
    step.setup_chain(rngs[rank], W, K)
    step.tune = bool(W)
    step.reset_tuning()
    point = points[rank]
    for i in range(W):
        if i == 0:
            step.iter_count = 0
        point, _ = step.step(point)
    step.stop_tuning()

    draws = np.empty((K, spec.n))
    tree = np.empty(K)
    step.profile(True)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        point, st = step.step(point)
        draws[i] = np.concatenate([point[k].ravel() for k in step.var_names])
        tree[i] = st[0]["tree_size"]
    barrier()
    dt = time.perf_counter() - t0
    dom_ms, dom_n, _ = step.profile_read()
    step.profile(False)

    # per-chain min bulk-ESS over a parameter subset (all hyper-parameters + random group effects)
    prng = np.random.default_rng(1)
    idx = np.unique(np.concatenate([np.arange(16), prng.choice(spec.n, size=min(args.ess_params, spec.n), replace=False)]))
    ess_vals = np.array([ess_bulk(draws[None, :, j]) for j in idx])
    min_ess = float(np.nanmin(ess_vals))
    leap = float(tree.sum())

    if dist is not None:
        t = torch.tensor([dt, min_ess, leap, dom_ms, float(dom_n)], dtype=torch.float64, device=comm_dev)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        allv = torch.stack(allt).cpu().numpy()
    else:
        allv = np.array([[dt, min_ess, leap, dom_ms, float(dom_n)]])
    if rank == 0:
        T = float(allv[:, 0].max())
        ess_total = float(allv[:, 1].sum())
        leap_total = float(allv[:, 2].sum())
        dom_avg_ms = float(allv[:, 3].sum() / max(allv[:, 4].sum(), 1))
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj) and args.rows_per_group == 4000 and args.groups == 1248:
            tr = json.load(open(tj))
            traffic, traffic_src = tr["k_rows_bytes_per_launch"], tr["source"]
        achieved = alg_bytes / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        out = {
            "metric": "effective samples/sec (and leapfrog steps/sec), 10k-param hierarchical logistic regression, one NUTS chain per GPU",
            "value": ess_total / T,
            "unit": "ESS/s (aggregate over chains; min-over-parameters bulk-ESS)",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * T / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"C2-{'L' if args.rows_per_group >= 1000 else 'S'} hier-logit-10k: G={args.groups} D=8 rows={N} n={spec.n}",
                "chains": chains,
                "parallelism": f"{world} independent chain(s), one per GPU, no data-path collective",
                "sampler": "NUTS target_accept=0.8 max_treedepth=10 init=jitter+adapt_diag",
            },
            "leapfrog_steps_per_sec": leap_total / T,
            "leapfrog_steps_per_sec_per_chain": [float(x) for x in (allv[:, 2] / allv[:, 0])],
            "ess_per_chain": [float(x) for x in allv[:, 1]],
            "mean_tree_size": leap_total / (K * world),
            "roofline": {
                "bound": "hbm",
                "kernel": "k_rows<8,2,4> (hierarchical-logit row pass)",
                "achieved": achieved,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": achieved / 8000.0,
                "frac_of_achievable_6.3TBps": achieved / 6300.0,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": dom_avg_ms,
                "launches_timed": int(allv[:, 4].sum()),
                "traffic": traffic,
                "traffic_source": traffic_src,
                # the whole leapfrog (row pass + kernel B + launch gaps + the per-draw / per-doubling host round trips of
                # the timed region) against the same line: SURVEY 8d bytes per leapfrog x leapfrogs/s per chain
                "leapfrog_algorithmic_bytes": alg_bytes + 144 * spec.n,
                "leapfrog_frac": (alg_bytes + 144 * spec.n) * (leap_total / T / world) / 8.0e12,
            },
        }
        if world == 1 and args.cpu_leapfrogs > 0:
            inv_mass = step._vector("var")
            eps = float(step._scalar("step_size"))
            out["cpu_baseline"] = cpu_baseline(spec, draws[-1], eps, inv_mass, args.cpu_leapfrogs, min_ess / max(leap, 1.0))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
