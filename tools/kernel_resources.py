"""Per-kernel register / scratch / LDS usage of the engine, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

usage: python tools/kernel_resources.py [log]   (without a log: compiles csrc/engine.hip into /tmp, ~2 min)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def remarks():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result", "-Wno-pass-failed",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pymc_amd", "csrc"), "-shared", "-fPIC",
           os.path.join(ROOT, "pymc_amd", "csrc", "engine.hip"), "-o", "/tmp/kernel_resources.so", "-Rpass-analysis=kernel-resource-usage"]
    return subprocess.run(cmd, capture_output=True, text=True).stderr


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[: len(names)]
    except FileNotFoundError:
        return names


def main():
    txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else remarks()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    rows = []
    for b in blocks:
        name = b.split("\n")[0].split(" ")[0]

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    names = demangle([r[0] for r in rows])
    print(f"{'kernel':100s} vgpr agpr sgpr scratch occ   lds")
    for nm, r in zip(names, rows):
        nm = re.sub(r"^void ", "", nm)
        nm = re.sub(r"\(.*$", "", nm)
        print(f"{nm[:100]:100s} {r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:7d} {r[5]:3d} {r[6]:5d}")


if __name__ == "__main__":
    main()
