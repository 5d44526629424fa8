"""The C-ABI library loads without a GPU and exports exactly what include/nuts_mi355.h declares.

No compute calls: on a GPU-less host the only behaviour exercised is that the
product path FAILS LOUDLY (there is no CPU fallback).
"""

import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nuts_mi355.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build_engine()
    from pymc_amd import _lib

    return _lib.load()


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(nuts_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_all_exported_and_bound(lib):
    from pymc_amd import _lib

    declared = _declared_functions()
    assert len(declared) >= 25
    assert sorted(_lib.SYMBOLS) == declared, "ctypes binding table and header disagree"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    missing = [n for n in declared if n not in exported]
    assert not missing, f"declared in the header but not exported: {missing}"
    extra = sorted(n for n in exported if n.startswith("nuts_") and n not in declared)
    assert not extra, f"exported but not declared in the header: {extra}"


def test_struct_layouts_match_the_c_header(tmp_path):
    """sizeof/offsetof of every ABI struct, compiled by gcc from the header, equals the ctypes mirror."""
    from pymc_amd import _lib

    structs = {
        "nuts_operand": _lib.Operand, "nuts_term": _lib.Term, "nuts_instr": _lib.Instr, "nuts_factor": _lib.Factor, "nuts_var": _lib.Var,
        "nuts_data_ref": _lib.DataRef, "nuts_model_spec": _lib.ModelSpecC, "nuts_chain_config": _lib.ChainConfig,
        "nuts_draw_stats": _lib.DrawStats, "nuts_hmc_stats": _lib.HmcStats, "nuts_lin": _lib.Lin,
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_default_config_matches_reference_defaults(lib):
    """BaseHMC.__init__ / NUTS.__init__ defaults (base_hmc.py:82-100, nuts.py:132-140)."""
    from pymc_amd import _lib

    cfg = _lib.ChainConfig()
    lib.nuts_chain_config_default(C.byref(cfg))
    assert (cfg.step_scale, cfg.Emax, cfg.target_accept) == (0.25, 1000.0, 0.8)
    assert (cfg.gamma, cfg.k, cfg.t0) == (0.05, 0.75, 10.0)
    assert (cfg.max_treedepth, cfg.early_max_treedepth, cfg.adapt_step_size) == (10, 8, 1)
    assert (cfg.adaptation_window, cfg.discard_window, cfg.adaptation_window_multiplier) == (101, 50, 1.0)


def test_no_cpu_fallback(lib):
    """Without a HIP device the product path must raise, never compute on the host."""
    if lib.nuts_device_count() > 0:
        pytest.skip("a GPU is visible: the loud-failure path is for GPU-less hosts")
    from pymc_amd import _lib, models
    from pymc_amd.value_grad import DeviceValueGradFunction

    with pytest.raises(_lib.EngineError, match="no HIP device|no CPU fallback"):
        DeviceValueGradFunction(models.eight_schools())


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pymc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "oracle/" not in txt or f.endswith(".md"), f


def test_streaming_kernels_hold_their_tiles_in_registers(lib, tmp_path):
    """The group-aligned row pass (csrc/rows_ga_kernel.h) issues its tile loads from inline assembly into registers the compiler
    does not know are pending: a spill of one of them would store a register whose load has not landed yet -- wrong numbers,
    no fault.  The code object's own metadata says whether the register allocator spilled: every `k_rows_ga` instantiation
    must report zero spilled VGPRs and no scratch.  (The span-partitioned `k_rows` uses ordinary loads: a spill there is correct
    and only costs time, so it is not part of this guard.)"""
    from pymc_amd import _lib

    tools = "/opt/rocm/lib/llvm/bin"
    needed = [os.path.join(tools, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in needed):
        pytest.skip("LLVM binutils of the ROCm toolchain not present")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "gfx950.co")
    subprocess.check_call([needed[0], f"--dump-section=.hip_fatbin={fat}", os.environ.get("PYMC_AMD_LIB", _lib.LIB_PATH), str(tmp_path / "copy.so")])
    subprocess.check_call([needed[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.check_output([needed[2], "--notes", co], text=True)
    kernels = {}
    for block in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        spills = re.search(r"\.vgpr_spill_count:\s+(\d+)", block)
        scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
        if name and spills and scratch:
            kernels[name.group(1)] = (int(spills.group(1)), int(scratch.group(1)))
    # (`k_rows_ga_multi`, the chain-group form, uses ordinary loads -- csrc/rows_ga_multi_kernel.h says why -- and is not part of this guard)
    ga = {k: v for k, v in kernels.items() if re.match(r"_Z\d+k_rows_gaI", k)}
    assert len(ga) >= 8, sorted(kernels)[:10]
    # the group-block pass (rows_gb_kernel.h) uses ordinary loads; it is held to the same bar because a spill inside its short
    # per-group loop would cost more than the loop itself (`k_vector`, which it replaces at C2-S, carries 144 B of scratch)
    # (of the merged launch `k_rows_gb_multi` the default register budget only: the 128- / 80-register instantiations exist for the
    # A/B that showed their spills cost more than the rounds they save)
    gb = {k: v for k, v in kernels.items() if re.match(r"_Z\d+k_rows_gbI", k) or re.match(r"_Z\d+k_rows_gb_multiILi\d+ELi\d+ELi2EE", k)}
    assert len(gb) >= 4, sorted(kernels)[:10]
    for name, (spills, scratch) in gb.items():
        assert spills == 0, (name, spills, scratch)
    # Both passes CALL the out-of-line auxiliary-workgroup function (csrc/rows_aux.h: the element-wise interpreter for whatever
    # the model has beyond the closed forms), whose register allocation -- spills included -- is its own: the kernels' scratch
    # size is the callee's.  What must hold for the kernels themselves is checked in the disassembly: every scratch instruction
    # of a `k_rows_ga` instantiation sits next to that call (the registers the calling convention saves around it), and the call
    # comes BEFORE the first hand-counted tile load -- nothing is stored or reloaded while such a load is in flight.
    objdump = os.path.join(tools, "llvm-objdump")
    if os.path.exists(objdump):
        for name in ga:
            dis = subprocess.check_output([objdump, "-d", f"--disassemble-symbols={name}", co], text=True).split("\n")
            ins = [l.split("//")[0].strip() for l in dis if l.startswith("\t")]
            calls = [i for i, l in enumerate(ins) if l.startswith("s_swappc_b64")]
            scr = [i for i, l in enumerate(ins) if l.startswith("scratch_")]
            tiles = [i for i, l in enumerate(ins) if re.match(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[", l)]
            assert tiles, name
            assert all(any(abs(i - c) <= 80 for c in calls) for i in scr), (name, scr[:5], calls)
            assert all(c < tiles[0] for c in calls) and all(i < tiles[0] for i in scr), (name, calls, tiles[0])
    else:
        for name, (spills, scratch) in ga.items():
            assert spills <= 1, (name, spills, scratch)
    # kernels that evaluate factors exist twice: with the expression-program interpreter (an out-of-line call with two 16-entry
    # scratch arrays) and without it; a model without programs must get the second -- with the call compiled in, k_small_draw<1024>
    # spilled 357 registers and ran 59 us per leapfrog instead of 24 at n = 1002 (tools/small_bench.py)
    for nt in (256, 512):
        lean = kernels[next(k for k in kernels if re.match(rf"_Z12k_small_drawILi{nt}ELb0E", k))]
        fat = kernels[next(k for k in kernels if re.match(rf"_Z12k_small_drawILi{nt}ELb1E", k))]
        assert lean == (0, 0) and fat[1] > 1000, (nt, lean, fat)
    # ... and none of the lean ones touches scratch: the per-element evaluator passes its arguments and partials by value
    # (dist_eval_v, model_dev.h); with pointer parameters they sat in 80 - 144 B of scratch per thread
    for pat in (r"_Z8k_vectorILi1ELb0E", r"_Z9k_controlILb0E"):
        assert kernels[next(k for k in kernels if re.match(pat, k))] == (0, 0), pat
