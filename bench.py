#!/usr/bin/env python
"""Benchmark of the NUTS hot path on MI355X (contract: see the round prompt, section 4).

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2-L hier-logit-10k"):
hierarchical logistic regression, n = 10 000 parameters (G=1248 groups x D=8),
N = 4 992 000 observation rows (X fp64 = 319.5 MB > 256 MiB LLC: HBM regime),
one NUTS chain per GPU, reference sampler defaults (target_accept 0.8,
max_treedepth 10 / 8 early, jitter+adapt_diag).  `--workload c3` runs BASELINE.json
configs[2] instead (MvNormal, full 2048 x 2048 covariance; cache-resident).

A "step" is one NUTS draw.  `--warmup W` tuning draws run untimed, then EXACTLY
`--steps K` post-warmup draws are timed between barrier + device synchronisation.

value: BASELINE.json's metric names two numbers, effective samples/sec and leapfrog
steps/sec.  Effective samples/sec = sum over chains of the min-over-ALL-parameters
bulk-ESS of the K timed draws / max-over-ranks wall time -- it is only an estimate of
anything when the chain is tuned and K is a few hundred draws, so it is `value` when
K >= 200 and W >= 200; for shorter runs (e.g. a driver smoke run with K = 20) `value`
is the aggregate leapfrog steps/sec and the ESS fields are null (`value_is` says
which).  `convergence` carries max R-hat, min / median ESS and WHICH parameter has
the minimum, so a small min-ESS can be traced to the coordinate that does not mix.

roofline: dominant kernel = the hierarchical-logit row pass (csrc/rows_kernel.h /
rows_ga_kernel.h; with NUTS_GA_VARIANT=32 the persistent tree kernel of
rows_ga_tree.h, which runs a whole NUTS tree per launch); algorithmic bytes per pass
= 69 B/row x N (SURVEY.md 8d B_model: X row + y + int32 group id), divided by the
average duration of one pass measured with HIP events on the library stream during
the timed region (1 launch in 8 is bracketed by events; launches that drain after
the tree terminated are included, exactly as in the rocprofv3 summary under
profiles/; a tree launch is bracketed whole and its duration divided by the leaves
the tree ran).  4 of the 69 bytes
are bytes the kernel AVOIDS (group structure is read as G+1 row pointers), so
`frac_traffic` = PMC bytes / duration / peak is reported next to `frac`.
`traffic` = HBM bytes per launch from the rocprofv3 PMC passes of
tools/gpu_round.sh (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
MI355X_MICROARCH.md), read from profiles/traffic.json together with the hash of the
kernel sources it was measured on (`traffic_build_matches`).

cpu_baseline (rank 0, N=1 only): the oracle's reference-order leapfrog
(`oracle/ref_sampler.Leapfrog` = the restatement that reproduces the reference's
CpuLeapfrogIntegrator bitwise) over (i) the single-threaded gcc -O3 -march=native
restatement of logp+grad (stand-in for PyTensor's C linker, which cannot run here),
one chain on one core; (ii) the same, 8 chains concurrently on 8 pinned cores (the
reference's own configuration at 8 chains, mcmc.py:1203-1224); (iii) the NumPy
restatement, one core.  A bounded number of leapfrogs of the same workload each;
kind = "port".  Its ESS/s is DERIVED (CPU leapfrog/s x the GPU run's ESS per
leapfrog: the sampler is the same algorithm, so ESS per leapfrog is shared) -- the
measured quantity is leapfrog steps/s.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ESS_MIN_DRAWS = 200


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3"], help="c2 = hier-logit-10k (the headline), c3 = MvNormal 2048")
    ap.add_argument("--rows-per-group", type=int, default=4000, help="4000 = C2-L (HBM regime), 80 = C2-S (cache resident)")
    ap.add_argument("--groups", type=int, default=1248)
    ap.add_argument("--mvn-k", type=int, default=2048)
    ap.add_argument("--seed", type=int, default=20160911)
    ap.add_argument("--cpu-leapfrogs", type=int, default=100, help="bounded CPU-baseline sample per leg (0 disables)")
    ap.add_argument("--draw-batch", type=int, default=int(os.environ.get("PYMC_AMD_DRAW_BATCH", "64")), help="post-tuning transitions per C call")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for tests)")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-file", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--share-gpu", action="store_true", help="TEST ONLY: every rank uses GPU 0 (exercises the N > 1 logic on a 1-GPU box)")
    return ap.parse_args()


def kernel_source_hash():
    h = hashlib.sha1()
    d = os.path.join(ROOT, "pymc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


# ---------------------------------------------------------------------------
# CPU baseline: reference-order leapfrogs on the host cores
# ---------------------------------------------------------------------------

def _cpu_leapfrogs(f, n, q, step_size, inv_mass, n_leap, core=None):
    from oracle import ref_sampler

    if core is not None:
        try:
            os.sched_setaffinity(0, {core})
        except Exception:
            pass
    pot = ref_sampler.DiagPotential(inv_mass)
    integ = ref_sampler.Leapfrog(pot, f)
    rng = np.random.default_rng(0 if core is None else core)
    s = integ.compute_state(np.array(q), rng.normal(size=n) / np.sqrt(inv_mass))
    t0 = time.perf_counter()
    for _ in range(n_leap):
        s = integ.step(step_size, s)
    return n_leap / (time.perf_counter() - t0)


def cpu_worker_main(args):
    """`bench.py --cpu-worker CORE --cpu-worker-file F`: one pinned process of the concurrent CPU leg (a fresh interpreter:
    nothing of the parent's HIP runtime is forked)."""
    from oracle import c_logit
    from pymc_amd import models

    try:
        import threadpoolctl

        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    k = np.load(args.cpu_worker_file)
    spec = models.hier_logit(G=args.groups, D=8, rows_per_group=args.rows_per_group, seed=20160911)
    f = c_logit.CHierLogit(spec, so_path=str(k["so"]))
    lps = _cpu_leapfrogs(f, spec.n, k["q"], float(k["eps"]), k["inv_mass"], int(k["n_leap"]), args.cpu_worker)
    print(json.dumps({"core": args.cpu_worker, "leapfrog_steps_per_sec": lps}), flush=True)


def cpu_baseline_c2(args, spec, q, step_size, inv_mass, n_leap, ess_per_leapfrog):
    import subprocess
    import tempfile

    from oracle import c_logit, ref_models

    try:
        import threadpoolctl

        threadpoolctl.threadpool_limits(1)
    except Exception:  # pragma: no cover
        pass
    so = c_logit.build_native()   # compiled for the box it is timed on
    t0 = time.perf_counter()
    lps_c1 = _cpu_leapfrogs(c_logit.CHierLogit(spec, so_path=so), spec.n, q, step_size, inv_mass, n_leap)
    # 8 chains concurrently, one process per chain pinned to one core (mcmc.py:1203-1224: blas_cores // cores = 1 thread each)
    ncore = min(8, os.cpu_count() or 1)
    n8 = max(10, n_leap // 3)
    per = []
    try:
        with tempfile.TemporaryDirectory() as td:
            wf = os.path.join(td, "w.npz")
            np.savez(wf, q=q, eps=step_size, inv_mass=inv_mass, n_leap=n8, so=so)
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(c), "--cpu-worker-file", wf,
                                       "--groups", str(args.groups), "--rows-per-group", str(args.rows_per_group)],
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for c in range(ncore)]
            for p in procs:
                o, _ = p.communicate(timeout=600)
                per.append(json.loads(o.strip().splitlines()[-1])["leapfrog_steps_per_sec"])
    except Exception as e:  # pragma: no cover
        print(f"bench: concurrent CPU leg failed: {e}", file=sys.stderr)
    # NumPy restatement (what the graph would cost without a C linker), a shorter sample: it is several times slower
    n_np = max(4, n_leap // 12)
    lps_np = _cpu_leapfrogs(ref_models.SpecLogpGrad(spec), spec.n, q, step_size, inv_mass, n_np)
    dt = time.perf_counter() - t0
    ess_derived = None if ess_per_leapfrog is None else lps_c1 * ess_per_leapfrog
    return {
        "value": lps_c1 if ess_derived is None else ess_derived,
        "unit": "leapfrog steps/s" if ess_derived is None else "ESS/s (derived: measured leapfrog steps/s x the GPU run's ESS per leapfrog)",
        "leapfrog_steps_per_sec": lps_c1,
        "ess_per_sec_derived": ess_derived,
        "cores": 1,
        "kind": "port",
        "host_cores_available": os.cpu_count(),
        "concurrent_8_chains": {
            "cores": len(per), "leapfrog_steps_per_sec_per_chain": [float(x) for x in per],
            "leapfrog_steps_per_sec_total": float(sum(per)), "leapfrogs_each": n8,
        },
        "numpy_variant": {"cores": 1, "leapfrog_steps_per_sec": lps_np, "leapfrogs": n_np},
        "sample": f"{n_leap} reference-order leapfrog steps of the same workload on one core (oracle integrator + single-threaded gcc -O3 "
        f"-march=native fused logp/grad, compiled on this box), then {n8} each in {len(per)} concurrent pinned processes, then {n_np} with the "
        f"NumPy logp/grad; {dt:.1f} s in all.  Restated reference CPU path (PyTensor unavailable)",
    }


def cpu_baseline_c3(spec, q, step_size, inv_mass, n_leap):
    from oracle import ref_models

    try:
        import threadpoolctl

        threadpoolctl.threadpool_limits(1)
    except Exception:  # pragma: no cover
        pass
    t0 = time.perf_counter()
    lps = _cpu_leapfrogs(ref_models.SpecLogpGrad(spec), spec.n, q, step_size, inv_mass, n_leap)
    return {
        "value": lps, "unit": "leapfrog steps/s (one chain on one host core)", "leapfrog_steps_per_sec": lps, "cores": 1, "kind": "port",
        "host_cores_available": os.cpu_count(),
        "sample": f"{n_leap} reference-order leapfrog steps (oracle integrator + SciPy Cholesky-solve logp/grad, BLAS threads = 1) in {time.perf_counter() - t0:.1f} s",
    }


# ---------------------------------------------------------------------------

def main():
    args = parse()
    if args.cpu_worker is not None:
        return cpu_worker_main(args)
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
    from pymc_amd.sampling import init_nuts, sample_draws
    from pymc_amd.step import get_random_generator
    from pymc_amd.stats import ess_bulk_many, rhat_many

    c3 = args.workload == "c3"
    if c3:
        spec = models.mvnormal(n=args.mvn_k)
        N = 0
    else:
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
    os.environ["PYMC_AMD_DRAW_BATCH"] = str(args.draw_batch)
    # the per-chain loop of `_iter_sample` (mcmc.py:1503-1583), split at the end of tuning so that exactly the K
    # post-warmup transitions sit between the two barriers
    step.setup_chain(rngs[rank], W, K)
    step.tune = bool(W)
    step.reset_tuning()
    point = points[rank]
    step.iter_count = 0
    _, _, point = sample_draws(step, point, W)
    step.stop_tuning()

    step.profile(True)
    barrier()
    t0 = time.perf_counter()
    draws, stats_list, point = sample_draws(step, point, K)
    barrier()
    dt = time.perf_counter() - t0
    dom_ms, dom_n, _ = step.profile_read()
    step.profile(False)
    tree = np.array([s["tree_size"] for s in stats_list])
    n_div = int(sum(bool(s["diverging"]) for s in stats_list))
    leap = float(tree.sum())

    ess_ok = K >= ESS_MIN_DRAWS and W >= ESS_MIN_DRAWS
    conv = None
    min_ess = float("nan")
    if ess_ok:
        ess = ess_bulk_many(draws[None])
        rh = rhat_many(draws[None])          # split R-hat of the one chain this rank ran
        j = int(np.nanargmin(ess))
        names = []
        for v in spec.vars:
            names += [f"{v.value_name}[{k}]" for k in range(v.size)]
        min_ess = float(ess[j])
        conv = {"min_ess": min_ess, "min_ess_param_index": j, "min_ess_param": names[j], "median_ess": float(np.median(ess)),
                "ess_5pct": float(np.percentile(ess, 5)), "rhat_max": float(np.nanmax(rh)), "rhat_max_param": names[int(np.nanargmax(rh))],
                "rhat_of_min_ess_param": float(rh[j]), "n_params": int(spec.n), "divergences": n_div}
        if not c3:
            # The non-centred parametrisation SURVEY 8 prescribes leaves (mu_d, mean_g z_{g,d}) on a ridge when every group has
            # thousands of rows: the combination the likelihood identifies, beta_bar_d = mu_d + sigma_d
            # mean_g z_{g,d}, is reported next to the per-coordinate minimum so that the two can be told apart.
            v = {x.name: x for x in spec.vars}
            D_ = v["mu"].size
            mu_ = draws[:, v["mu"].offset : v["mu"].offset + D_]
            sg_ = np.exp(draws[:, v["sigma"].offset : v["sigma"].offset + D_])
            zbar = draws[:, v["z"].offset : v["z"].offset + v["z"].size].reshape(len(draws), -1, D_).mean(axis=1)
            eb = ess_bulk_many((mu_ + sg_ * zbar)[None])
            conv["identified_combination"] = {"what": "beta_bar_d = mu_d + sigma_d * mean_g z[g, d]", "min_ess": float(eb.min()),
                                              "median_ess": float(np.median(eb)), "corr_mu_zbar_d0": float(np.corrcoef(mu_[:, 0], zbar[:, 0])[0, 1])}

    vec = [dt, min_ess if ess_ok else 0.0, leap, dom_ms, float(dom_n)]
    if dist is not None:
        t = torch.tensor(vec, dtype=torch.float64, device=comm_dev)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        allv = torch.stack(allt).cpu().numpy()
        convs = [None] * world
        dist.all_gather_object(convs, conv)
    else:
        allv = np.array([vec])
        convs = [conv]
    if rank == 0:
        T = float(allv[:, 0].max())
        leap_total = float(allv[:, 2].sum())
        lps_total = leap_total / T
        dom_avg_ms = float(allv[:, 3].sum() / max(allv[:, 4].sum(), 1))
        ess_total = float(allv[:, 1].sum()) if ess_ok else None
        achieved = alg_bytes / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        traffic, traffic_src, traffic_match = None, None, None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if not c3 and os.path.exists(tj) and args.rows_per_group == 4000 and args.groups == 1248:
            tr = json.load(open(tj))
            traffic, traffic_src = tr["k_rows_bytes_per_launch"], tr["source"]
            traffic_match = tr.get("kernel_source_hash") == kernel_source_hash()
        tj3 = os.path.join(ROOT, "profiles", "traffic_c3.json")
        if c3 and os.path.exists(tj3) and args.mvn_k == 2048:
            tr = json.load(open(tj3))
            traffic, traffic_src = tr["k_mvn_aligned_bytes_per_launch"], tr["source"]
            traffic_match = tr.get("kernel_source_hash") == kernel_source_hash()
        leap_bytes = alg_bytes + 144 * spec.n
        if c3:
            workload = f"C3 mvn-{args.mvn_k}: MvNormal, full {args.mvn_k}x{args.mvn_k} covariance, n={spec.n}"
            aligned = int(step._logp_dlogp_func.model_scalar("mvn_row_aligned"))
            kernel = (f"k_mvn_aligned<{aligned}> (precision mat-vec whose workgroups also finish the leapfrog: one launch per leapfrog"
                      if aligned else "k_mvn_matvec (precision mat-vec") + \
                "; cache-resident: 33.5 MB < 256 MiB Infinity Cache -- the HBM line does not bound it)"
        else:
            workload = f"C2-{'L' if args.rows_per_group >= 1000 else 'S'} hier-logit-10k: G={args.groups} D=8 rows={N} n={spec.n}"
            kernel = "hierarchical-logit row pass (k_rows_ga / k_rows)"
        out = {
            "metric": "effective samples/sec (and leapfrog steps/sec), 10k-param hierarchical logistic regression, one NUTS chain per GPU"
            if not c3 else "leapfrog steps/sec (and effective samples/sec), MvNormal 2048, one NUTS chain per GPU",
            "value": (ess_total / T) if ess_ok else lps_total,
            "unit": "ESS/s (aggregate over chains; min-over-all-parameters bulk-ESS)" if ess_ok else "leapfrog steps/s (aggregate over chains)",
            "value_is": "ess_per_sec" if ess_ok else f"leapfrog_steps_per_sec (ESS needs steps >= {ESS_MIN_DRAWS} and warmup >= {ESS_MIN_DRAWS})",
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
                "workload": workload,
                "chains": chains,
                "parallelism": f"{world} independent chain(s), one per GPU, no data-path collective",
                "sampler": "NUTS target_accept=0.8 max_treedepth=10 init=jitter+adapt_diag",
            },
            "schedule": ("persistent tree kernel: one launch per NUTS tree (csrc/rows_ga_tree.h)" if step._scalar("tree_kernel") else
                         "group-aligned row pass: one launch per leapfrog, control work folded into the next row pass, also across doublings "
                         "(csrc/rows_ga_kernel.h)" if (not c3 and step._logp_dlogp_func.model_scalar("rows_group_aligned")) else
                         "row-aligned MvNormal pass: one launch per leapfrog, control work folded into the next launch, also across doublings "
                         "(csrc/kernels.h, k_mvn_aligned)" if (c3 and step._logp_dlogp_func.model_scalar("mvn_row_aligned")) else
                         "two launches per leapfrog (data pass + O(n) kernel), control work folded into the next data pass (csrc/kernels.h)"),
            "leapfrog_steps_per_sec": lps_total,
            "leapfrog_steps_per_sec_per_chain": [float(x) for x in (allv[:, 2] / allv[:, 0])],
            "ess_per_sec": (ess_total / T) if ess_ok else None,
            "ess_per_chain": [float(x) for x in allv[:, 1]] if ess_ok else None,
            "convergence": convs if ess_ok else None,
            "mean_tree_size": leap_total / (K * world),
            "roofline": {
                "bound": "hbm",
                "kernel": kernel,
                "achieved": achieved,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": achieved / 8000.0,
                "frac_of_achievable_6.3TBps": achieved / 6300.0,
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_note": None if c3 else "69 B/row = 64 (X) + 1 (y) + 4 (int32 group id); the kernel reads G+1 row pointers instead of the "
                "group ids, so 4 of the 69 are bytes it avoids -- frac_traffic prices the bytes actually moved",
                "avg_launch_ms": dom_avg_ms,
                "launches_timed": int(allv[:, 4].sum()),   # (passes over the data covered by the bracketed launches)
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_build_matches": traffic_match,
                "frac_traffic": (traffic / (dom_avg_ms * 1e-3) / 8.0e12) if (traffic and dom_avg_ms > 0) else None,
                # the whole leapfrog (row pass + O(n) work + launch gaps + the per-draw / per-doubling host round trips of
                # the timed region) against the same line: SURVEY 8d bytes per leapfrog x leapfrogs/s per chain
                "leapfrog_algorithmic_bytes": leap_bytes,
                "leapfrog_frac": leap_bytes * (lps_total / world) / 8.0e12,
            },
        }
        if world == 1 and args.cpu_leapfrogs > 0:
            inv_mass = step._vector("var")
            eps = float(step._scalar("step_size"))
            if c3:
                out["cpu_baseline"] = cpu_baseline_c3(spec, draws[-1], eps, inv_mass, args.cpu_leapfrogs)
            else:
                out["cpu_baseline"] = cpu_baseline_c2(args, spec, draws[-1], eps, inv_mass, args.cpu_leapfrogs, (min_ess / max(leap, 1.0)) if ess_ok else None)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
