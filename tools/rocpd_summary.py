#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (SQLite) outputs into small text files for profiles/.

usage: rocpd_summary.py <kernel-trace .db> [--pmc <pmc .db> ...] > profiles/<name>.txt

Kernel table = what `rocprofv3 --kernel-trace --stats` reports (count / total / avg / min / max per
kernel).  PMC tables: average counter value per kernel; FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950
FETCH_SIZE counts 128-B requests of a wide coalesced read as 64 B (MI355X_MICROARCH.md, HBM section):
the corrected read traffic is 2 x FETCH_SIZE.
"""
import sqlite3
import sys


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
    med = {}
    for name, dur in cur.execute("select name, end-start from kernels"):
        med.setdefault(name, []).append(dur)
    med = {k: sorted(v)[len(v) // 2] for k, v in med.items()}
    tot = sum(r[2] for r in rows) or 1
    t0, t1 = list(cur.execute("select min(start), max(end) from kernels"))[0]
    print(f"# kernel trace: {path}")
    print(f"# GPU busy {tot/1e6:.3f} ms over a {(t1-t0)/1e6:.3f} ms span ({100*tot/(t1-t0):.1f}% busy)")
    print(f"{'kernel':70s} {'calls':>8s} {'total_ms':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'median_us':>10s}")
    for r in rows:
        print(f"{r[0][:70]:70s} {r[1]:8d} {r[2]/1e6:11.3f} {r[3]/1e3:9.2f} {r[4]/1e3:9.2f} {r[5]/1e3:9.2f} {100*r[2]/tot:6.2f} {med[r[0]]/1e3:10.2f}")


def gap_stats(path):
    """Idle time on the GPU before each kernel (start - end of the previous dispatch), by kernel."""
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    agg = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        agg.setdefault(n1, []).append(max(0, s1 - e0))
    print("\n# idle gap before each kernel (us): mean and median over dispatches")
    for n, g in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        g = sorted(g)
        print(f"{n[:70]:70s} {len(g):8d} {sum(g)/len(g)/1e3:9.2f} {g[len(g)//2]/1e3:9.2f}")


def pmc_stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute(
        "select kernel_name, counter_name, count(*), avg(value), max(value) from counters_collection group by kernel_name, counter_name order by 4 desc"))
    print(f"\n# pmc: {path}")
    print(f"{'kernel':70s} {'counter':>14s} {'n':>6s} {'avg':>14s} {'max':>14s}")
    for r in rows:
        print(f"{r[0][:70]:70s} {r[1]:>14s} {r[2]:6d} {r[3]:14.3f} {r[4]:14.3f}")


def traffic_json(fetch_db, write_db, out_path, kernel_like, tag, src_hash, key="k_rows_bytes_per_launch", note=""):
    """HBM bytes per launch of the dominant kernel from the two PMC passes, with the guide's gfx950 correction: read side =
    2 x FETCH_SIZE (KiB), write side = WRITE_SIZE (KiB).  The MEAN over FULL launches: launches queued behind a tree that had
    already terminated drain as no-ops (they read a flag and leave) and are told apart by their counter value -- less than half of
    the largest one -- instead of being averaged in (their share is reported)."""
    import json

    def stats(path, counter):
        cur = sqlite3.connect(path).cursor()
        vals = [r[0] for r in cur.execute("select value from counters_collection where counter_name = ? and kernel_name like ?", (counter, f"%{kernel_like}%"))]
        if not vals:
            return None
        top = max(vals)
        full = [v for v in vals if v >= 0.5 * top]
        return {"max": top, "mean_all": sum(vals) / len(vals), "mean_full": sum(full) / len(full), "n": len(vals), "n_full": len(full)}

    f, w = stats(fetch_db, "FETCH_SIZE"), stats(write_db, "WRITE_SIZE")
    if f is None or w is None:
        return
    # (the write counter of a drained launch is not far from a full one's -- a few KB either way -- so the split is made on the read side)
    out = {
        key: int(2 * f["mean_full"] * 1024 + w["mean_all"] * 1024),
        "fetch_size_kib_mean_full_launches": f["mean_full"], "fetch_size_kib_max": f["max"], "fetch_size_kib_mean_all": f["mean_all"],
        "write_size_kib_mean": w["mean_all"], "write_size_kib_max": w["max"],
        "launches": f["n"], "full_launches": f["n_full"], "drained_launches": f["n"] - f["n_full"],
        "kernel": kernel_like, "kernel_source_hash": src_hash,
        "source": f"profiles/{tag}_profile.txt: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), MEAN over the full launches "
                  "(drained launches, recognised by a read counter below half the maximum, are left out); read side doubled per MI355X_MICROARCH.md "
                  "(gfx950 counts 128-B requests as 64 B)" + note,
    }
    json.dump(out, open(out_path, "w"), indent=1)


def draw_timeline(path, tail_frac=0.5):
    """Per-draw anatomy of the LAST `tail_frac` of the trace (the timed, post-warm-up part of a bench run): a draw = the dispatches
    from one k_draw_start to the next; time inside the row-pass / data-pass kernels, inside the other kernels, and idle, by what
    the GPU was waiting to start."""
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    if not rows:
        return
    t_lo = rows[0][1] + (1.0 - tail_frac) * (rows[-1][2] - rows[0][1])
    rows = [r for r in rows if r[1] >= t_lo]
    starts = [i for i, r in enumerate(rows) if r[0].startswith("k_draw_start")]
    if len(starts) < 3:
        return
    tot = {"draws": 0, "span": 0.0, "dominant": 0.0, "other_kernels": 0.0, "idle": 0.0}
    idle_by, other_by = {}, {}
    for a, b in zip(starts, starts[1:]):
        seg = rows[a:b + 1]
        tot["draws"] += 1
        tot["span"] += seg[-1][1] - seg[0][1]
        for (n0, s0, e0), (n1, s1, e1) in zip(seg, seg[1:]):
            dom = "k_rows" in n0 or "k_mvn_aligned" in n0 or "k_tree_ga" in n0
            tot["dominant" if dom else "other_kernels"] += e0 - s0
            if not dom:
                other_by[n0[:40]] = other_by.get(n0[:40], 0.0) + (e0 - s0)
            g = max(0, s1 - e0)
            tot["idle"] += g
            idle_by[n1[:40]] = idle_by.get(n1[:40], 0.0) + g
    n = tot["draws"]
    print(f"\n# per-draw anatomy over the last {int(100 * tail_frac)} % of the trace ({n} draws): us per draw")
    print(f"span {tot['span']/n/1e3:9.2f}   dominant kernels {tot['dominant']/n/1e3:9.2f}   other kernels {tot['other_kernels']/n/1e3:8.2f}   idle {tot['idle']/n/1e3:8.2f}")
    print("other kernels:  " + ", ".join(f"{k} {v/n/1e3:.2f}" for k, v in sorted(other_by.items(), key=lambda kv: -kv[1])))
    print("idle before:    " + ", ".join(f"{k} {v/n/1e3:.2f}" for k, v in sorted(idle_by.items(), key=lambda kv: -kv[1])))


def launch_positions(path, tail_frac=0.5):
    """Mean duration of the dominant kernel by its POSITION in the draw, over the draws of the most common launch count (the
    tuned tree): position p is leaf j of doubling d in queue order, and -- with the control work folded into the next launch --
    carries the control work of position p - 1, whose merge depth is ctz(~j) of THAT leaf."""
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    if not rows:
        return
    t_lo = rows[0][1] + (1.0 - tail_frac) * (rows[-1][2] - rows[0][1])
    rows = [r for r in rows if r[1] >= t_lo]
    starts = [i for i, r in enumerate(rows) if r[0].startswith("k_draw_start")]
    draws = []
    for a, b in zip(starts, starts[1:]):
        seg = [(e - s0) / 1e3 for n, s0, e in rows[a:b] if "k_rows" in n or "k_mvn_aligned" in n]
        draws.append(seg)
    if not draws:
        return
    from collections import Counter
    L = Counter(len(d) for d in draws).most_common(1)[0][0]
    sel = [d for d in draws if len(d) == L]
    print(f"\n# dominant kernel by position in the draw ({len(sel)} draws of {L} launches): mean us  [doubling d, leaf j, merge depth of the control work it carries]")
    pos = []
    d = 0
    while len(pos) < L:
        for j in range(1 << d):
            pos.append((d, j))
        d += 1
    pos = pos[:L]
    by_m = {}
    for p in range(L):
        mean = sum(x[p] for x in sel) / len(sel)
        if p == 0:
            carried = "-"
        else:
            dp, jp = pos[p - 1]
            m = 0
            while (jp >> m) & 1 and m < dp:
                m += 1
            carried = f"m={m}" + (" last" if jp + 1 == (1 << dp) else "")
            by_m.setdefault(carried, []).append(mean)
        print(f"  p={p:3d} d={pos[p][0]} j={pos[p][1]:3d} carries {carried:10s} {mean:7.2f}")
    print("# by carried control work: " + ", ".join(f"{k}: {sum(v)/len(v):.2f} us (n={len(v)})" for k, v in sorted(by_m.items())))


def gap_positions(path, tail_frac=0.5):
    """Idle gap before the dominant kernel: quantiles over all its dispatches in the tail of the trace, and mean / median by POSITION
    in the draw (same draws as launch_positions).  A gap every launch pays (CP dispatch, cache maintenance at the kernel boundary)
    shows as a flat floor; one the host causes (a doubling queued late) shows at the first leaf of a doubling only."""
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    if not rows:
        return
    t_lo = rows[0][1] + (1.0 - tail_frac) * (rows[-1][2] - rows[0][1])
    rows = [r for r in rows if r[1] >= t_lo]
    dom = lambda n: "k_rows" in n or "k_mvn_aligned" in n
    gaps_all = sorted(max(0, s1 - e0) / 1e3 for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]) if dom(n1) and dom(n0))
    if not gaps_all:
        return
    q = lambda f: gaps_all[min(len(gaps_all) - 1, int(f * len(gaps_all)))]
    print(f"\n# idle gap between two consecutive dominant launches (us, {len(gaps_all)} gaps): min {gaps_all[0]:.2f}  p10 {q(0.1):.2f}  p50 {q(0.5):.2f}  "
          f"p90 {q(0.9):.2f}  p99 {q(0.99):.2f}  max {gaps_all[-1]:.2f}  mean {sum(gaps_all)/len(gaps_all):.2f}")
    starts = [i for i, r in enumerate(rows) if r[0].startswith("k_draw_start")]
    draws = []
    for a, b in zip(starts, starts[1:]):
        seg = rows[a:b]
        g = [max(0, seg[i][1] - seg[i - 1][2]) / 1e3 for i in range(1, len(seg)) if dom(seg[i][0])]
        draws.append(g)
    if not draws:
        return
    from collections import Counter
    L = Counter(len(d) for d in draws).most_common(1)[0][0]
    sel = [d for d in draws if len(d) == L]
    pos = []
    d = 0
    while len(pos) < L:
        for j in range(1 << d):
            pos.append((d, j))
        d += 1
    first, inner = [], []
    for p in range(L):
        col = sorted(x[p] for x in sel)
        (first if pos[p][1] == 0 else inner).append((sum(col) / len(col), col[len(col) // 2]))
    fm = lambda v, k: sum(x[k] for x in v) / max(len(v), 1)
    print(f"# by position ({len(sel)} draws of {L} launches): first leaf of a doubling mean {fm(first, 0):.2f} median {fm(first, 1):.2f} us (n={len(first)});  "
          f"inner leaves mean {fm(inner, 0):.2f} median {fm(inner, 1):.2f} us (n={len(inner)})")


def launch_time_json(trace_db, out_path, kernel_like, tag, src_hash, command=""):
    """Duration of the dominant kernel's launches from a `rocprofv3 --kernel-trace` run (no counters, no marker packets): median and
    mean over FULL launches (launches that drain behind a finished tree are told apart by their duration, less than half the
    median).  bench.py carries these next to its own HIP-event figure (whose marker packets inflate the bracketed launch)."""
    import json

    cur = sqlite3.connect(trace_db).cursor()
    dur = sorted(d for (d,) in cur.execute("select end-start from kernels where name like ?", (f"%{kernel_like}%",)))
    if not dur:
        raise SystemExit(f"no kernel like {kernel_like!r} in {trace_db}")
    med_all = dur[len(dur) // 2]
    full = [d for d in dur if d >= 0.5 * med_all]
    out = {"kernel_like": kernel_like, "launches": len(dur), "full_launches": len(full), "median_us": full[len(full) // 2] / 1e3,
           "mean_us": sum(full) / len(full) / 1e3, "mean_us_all_launches": sum(dur) / len(dur) / 1e3,
           "source": f"rocprofv3 --kernel-trace --stats -- {command}".strip(" -") + f" (tag {tag})", "kernel_source_hash": src_hash}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    args = sys.argv[1:]
    if args[0] == "--launch-time":   # --launch-time <trace.db> <out.json> <kernel substring> <tag> <source hash> [command]
        launch_time_json(*args[1:7])
        sys.exit(0)
    if args[0] == "--traffic":   # --traffic <fetch.db> <write.db> <out.json> <kernel substring> <tag> <source hash> [json key] [note]
        traffic_json(*args[1:9])
        sys.exit(0)
    kernel_stats(args[0])
    gap_stats(args[0])
    draw_timeline(args[0])
    launch_positions(args[0])
    gap_positions(args[0])
    rest = args[1:]
    for a in rest:
        if a != "--pmc":
            pmc_stats(a)

