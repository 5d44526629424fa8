#!/usr/bin/env python
"""Basic blocks of one kernel in hipcc's `-S` output, with instruction counts by class -- which block is the streaming loop, and what
is it made of?   usage: python tools/isa_blocks.py file.s <kernel-name-substring> [min instructions per block]"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_rcp_f64", "v_ldexp_f64", "v_rndne_f64", "v_max_f64", "v_min_f64", "v_cvt", "v_cmp", "v_div", "v_frexp", "v_trig", "v_floor_f64")) or "_f64" in op:
        return "fp64"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_mov", "v_accvgpr", "v_cndmask", "v_readlane", "v_writelane", "v_readfirstlane", "v_perm", "v_bfe", "v_and", "v_or", "v_lshl", "v_lshr", "v_add_u", "v_add_co", "v_sub", "v_mad_u", "v_xor", "v_not", "v_ashr", "v_mul_lo", "v_mul_hi", "v_add3", "v_lshl_add", "v_bfi", "v_alignbit", "v_pk", "v_med", "v_max_", "v_min_", "v_add_i", "v_addc")):
        return "valu-other"
    if op.startswith("v_"):
        return "valu-other"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(name) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, label = [], Counter(), "entry"
    order = []
    for l in lines[start + 1 : end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((label, cur)); order.append(label)
            cur, label = Counter(), m.group(1)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        cur[classify(op)] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            cur["->" + t.split()[-1]] += 0
            cur.setdefault("targets", [])
        if op.startswith(("s_cbranch", "s_branch")):
            cur["targets"] = cur.get("targets", []) + [t.split()[1]] if isinstance(cur.get("targets"), list) else [t.split()[1]]
    blocks.append((label, cur))
    pos = {lab: i for i, (lab, _) in enumerate(blocks)}
    tot = Counter()
    for lab, c in blocks:
        n = sum(v for k, v in c.items() if isinstance(v, int))
        for k, v in c.items():
            if isinstance(v, int):
                tot[k] += v
        back = [t for t in c.get("targets", []) if isinstance(c.get("targets"), list) and t in pos and pos[t] <= pos[lab]]
        if n >= floor or back:
            print(f"{lab:>12} {n:5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items()) if isinstance(v, int) and v) + (f"   LOOP-> {back}" if back else ""))
    print("total", sum(tot.values()), dict(tot))


if __name__ == "__main__":
    main()
