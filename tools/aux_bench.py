"""Throughput of the two BASELINE.json workloads next to the NUTS path, on one GPU:

  --workload c4   configs[3]: GLM, N = 1 M observations x P = 512 covariates, full-rank ADVI on minibatches of 1024 rows
                  -> optimisation steps per second (one step = `nuts_advi_steps`' two launches: the gathered rows, then the
                     row-aligned update that also finishes d logp / dz and forms the next step's z), with the bytes a step has to move and the CPU oracle's
                     steps per second beside it;
  --workload c5   configs[4]: Normal mixture, N = 100 k latent assignments + K component means:
                  `CompoundStep([NUTS(mu), CategoricalGibbsMetropolis(c)])` iterations per second, the Gibbs sweep alone, and the
                  CPU oracle's sweep beside it.

Prints ONE JSON line per workload (same spirit as bench.py: synthetic data of the named shape, inputs resident before the
timed region, HIP work synchronised by the C calls themselves).  These are reported measurements of the widened rows, not
the headline metric -- that is bench.py's."""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench_c4(args):
    from pymc_amd import models
    from pymc_amd.variational import FullRankADVI

    t0 = time.perf_counter()
    m = models.glm(N=args.glm_rows, P=args.glm_cols, batch_size=args.batch, seed=4)
    t_build = time.perf_counter() - t0
    inf = FullRankADVI(model=m, random_seed=2, device=0)
    idx, z0 = inf.draw_inputs(args.warmup)
    t0 = time.perf_counter()
    inf.run_steps(idx, z0)                       # uploads X (host -> HBM) and compiles nothing: the first call pays the copy
    t_first = time.perf_counter() - t0
    idx, z0 = inf.draw_inputs(args.steps)
    t0 = time.perf_counter()
    loss = inf.run_steps(idx, z0)
    T = time.perf_counter() - t0
    P, B, n_win = m.n, m.batch_size, 10
    n_l = P * (P + 1) // 2
    # bytes one step has to move (fp64): the B gathered rows of X + y, L read for z and written / read for the gradient and
    # the update, mu, and the adagrad_window ring (n_win slots of squared gradients: one written, all summed)
    step_bytes = 8 * (B * (P + 1) + 3 * n_l + 3 * P + (n_win + 1) * (n_l + P))
    out = {
        "workload": f"C4 glm-advi: N={m.X.shape[0]} P={P}, full-rank ADVI, minibatch {B}, adagrad_window(n_win={n_win})",
        "metric": "ADVI optimisation steps/sec", "value": args.steps / T, "unit": "steps/s", "steps": args.steps, "warmup": args.warmup,
        "us_per_step": 1e6 * T / args.steps, "launches_per_step": 2, "dtype": "f64", "data": "synthetic",
        "step_bytes_algorithmic": step_bytes, "achieved_GBps": step_bytes * args.steps / T / 1e9,
        "note": "latency-bound: two dependent launches per step move ~%.1f MB; X (%.1f GB) stays resident in HBM, only the drawn rows are read"
                % (step_bytes / 1e6, m.X.nbytes / 1e9),
        "final_loss": float(loss[-1]), "host_build_s": t_build, "first_call_s_incl_upload": t_first,
    }
    if args.cpu_steps > 0:
        from oracle import ref_advi

        glm, st = ref_advi.GLM(m.X, m.y, m.family, m.sigma, m.prior_sd), ref_advi.FullRankState(m.n)
        idx, z0 = inf.draw_inputs(args.cpu_steps)
        t0 = time.perf_counter()
        for k in range(args.cpu_steps):
            ref_advi.advi_step(glm, st, idx[k], z0[k])
        Tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": args.cpu_steps / Tc, "unit": "steps/s", "cores": "NumPy/BLAS threads of this box", "kind": "port",
                               "sample": f"{args.cpu_steps} steps of the NumPy oracle (oracle/ref_advi.py) on inputs of the same shape"}
    inf.close()
    print(json.dumps(out))


def bench_c5(args):
    from pymc_amd import models
    from pymc_amd.compound import CompoundStep
    from pymc_amd.gibbs import CategoricalGibbsMetropolis
    from pymc_amd.step import NUTS

    # --bayes: the fully Bayesian mixture -- Dirichlet weights and component scales are NUTS variables as well (mixture node given
    # the assignments; models.normal_mixture_bayes), the Gibbs step reads them from the point
    spec = models.normal_mixture_bayes(N=args.mix_n, K=3, seed=7) if args.bayes else models.normal_mixture(N=args.mix_n, K=3, seed=7)
    link = spec.mixture
    nuts = NUTS(model=spec, rng=1, device=0)
    gibbs = CategoricalGibbsMetropolis(model=spec, rng=2, device=0)
    comp = CompoundStep([nuts, gibbs])
    comp.setup_chain(np.random.default_rng(99), args.warmup, args.steps)
    c0 = np.random.default_rng(3).integers(0, 3, size=args.mix_n)
    point = {"mu": np.array([-1.0, 0.2, 1.5]), "c": c0.copy()}
    if args.bayes:
        point.update({"w_simplex__": np.zeros(2), "sigma_log__": np.zeros(3), "mu": np.array([-3.0, 0.2, 3.5])})
    nuts.tune = True
    nuts.reset_tuning()
    nuts.iter_count = 0
    for _ in range(args.warmup):
        point, _ = comp.step(point)
    comp.stop_tuning()
    prof = None
    if args.profile_host:
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    trees = 0
    for _ in range(args.steps):
        point, st = comp.step(point)
        trees += int(st[0]["tree_size"])
    T = time.perf_counter() - t0
    if prof is not None:
        import pstats
        import sys

        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(18)
    # the NUTS step alone (the assignments fixed), and what the plan look-ahead can supply when nothing else runs
    t0 = time.perf_counter()
    for _ in range(args.steps):
        point, _ = nuts.step(point)
    Tn = time.perf_counter() - t0
    from pymc_amd import gibbs as G

    pipe = G._PlanPipeline(np.random.default_rng(4).bit_generator.state, np.arange(args.mix_n, dtype="int32"), np.full(args.mix_n, 3, dtype="int32"), True)
    for _ in range(5):
        pipe.take()
    t0 = time.perf_counter()
    for _ in range(200):
        pipe.take()
    Tp = (time.perf_counter() - t0) / 200
    pipe.close()
    # the sweep alone
    t0 = time.perf_counter()
    for _ in range(args.steps):
        point, _ = gibbs.step(point)
    Tg = time.perf_counter() - t0
    N = args.mix_n
    out = {
        "workload": (f"C5 mixture, fully Bayesian: N={N} latent assignments, K=3; CompoundStep([NUTS(w ~ Dirichlet, mu, sigma), CategoricalGibbsMetropolis(c)])"
                     if args.bayes else
                     f"C5 mixture: N={N} latent assignments, K=3 component means; CompoundStep([NUTS(mu), CategoricalGibbsMetropolis(c)])"),
        "metric": "compound iterations/sec", "value": args.steps / T, "unit": "iterations/s", "steps": args.steps, "warmup": args.warmup,
        "ms_per_iteration": 1e3 * T / args.steps, "mean_tree_size": trees / args.steps,
        "gibbs_sweep_ms": 1e3 * Tg / args.steps, "gibbs_elements_per_sec": N * args.steps / Tg,
        "nuts_step_ms": 1e3 * Tn / args.steps, "plan_lookahead_ms_per_plan": 1e3 * Tp,
        "sweep_bytes_algorithmic": N * (8 + 8 + 8 + 8 + 8 + 8),
        "note": "a sweep = one device launch of N acceptance tests on a plan (the reference's PCG64 stream: shuffle, choice(k-1), uniform) that "
                "host threads replayed and uploaded while earlier sweeps ran (pymc_amd/gibbs.py _PlanPipeline); the sweep's own thread uploads "
                "3 K parameters and reads the assignments back.  plan_lookahead_ms_per_plan = what those threads can supply on this host",
        "dtype": "f64 / int64", "data": "synthetic",
    }
    if args.cpu_steps > 0:
        from oracle import ref_gibbs

        og = ref_gibbs.RefCategoricalGibbs(link.y, link.log_w_at(point), link.sigma_at(point), np.random.default_rng(5))
        c = c0.copy()
        k = max(1, min(args.cpu_steps, 3))
        t0 = time.perf_counter()
        for _ in range(k):
            c, _ = og.sweep(c, point["mu"])
        Tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / Tc, "unit": "sweeps/s", "cores": 1, "kind": "port",
                               "sample": f"{k} sweep(s) of the oracle restatement (oracle/ref_gibbs.py: per-element deltas, O(N) per sweep -- "
                                         "the reference's own sweep evaluates the full model per element, O(N^2))"}
    nuts.close(); gibbs.close()
    print(json.dumps(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["c4", "c5"], required=True)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--glm-rows", type=int, default=1_000_000)
    ap.add_argument("--glm-cols", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--mix-n", type=int, default=100_000)
    ap.add_argument("--profile-host", action="store_true", help="c5: cProfile of the timed loop's host thread, to stderr")
    ap.add_argument("--bayes", action="store_true", help="c5: weights (Dirichlet) and component scales are NUTS variables too")
    a = ap.parse_args()
    (bench_c4 if a.workload == "c4" else bench_c5)(a)
