"""Where a kernel's register spills execute: hipcc -S of a translation unit, then per kernel the spill instructions (scalar spills =
v_writelane / v_readlane on the spill VGPRs, vector spills = scratch_store / scratch_load) counted per loop depth (from the
compiler's own `Loop Header: Depth=N` block comments).  A spill outside every loop runs once per wavefront; one at depth 2-3 runs
hundreds of times.  usage: python scripts/spill_weight.py [kernels.hip] [-D...] [--kernels rescore_kernelILb0ELb0 prelim_kernelILb1ELb0ELb0]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm_of(src, extra):
    out = tempfile.mktemp(suffix=".s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "--cuda-device-only", "-S", *extra, src, "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    text = open(out).read().split("\n")
    os.unlink(out)
    return text


def analyse(lines, pat):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*" + pat + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    depth = 0
    by_depth = {}
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            c = m.group(2) or ""
            d = re.search(r"Depth=(\d+)", c)
            depth = int(d.group(1)) if d else 0
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        row = by_depth.setdefault(depth, dict(instr=0, valu=0, sgpr_reload=0, sgpr_spill=0, vgpr_reload=0, vgpr_spill=0))
        row["instr"] += 1
        row["valu"] += op.startswith("v_")
        row["sgpr_reload"] += op == "v_readlane_b32" and "; " not in t  # (reloads of spilled scalars; user readlanes carry no comment either)
        row["sgpr_spill"] += op == "v_writelane_b32"
        row["vgpr_reload"] += op.startswith("scratch_load")
        row["vgpr_spill"] += op.startswith("scratch_store")
    return by_depth


if __name__ == "__main__":
    args = sys.argv[1:]
    kernels = ["rescore_kernelILb0ELb0", "prelim_kernelILb1ELb0ELb0"]
    if "--kernels" in args:
        i = args.index("--kernels")
        kernels = args[i + 1:]
        args = args[:i]
    src = next((a for a in args if not a.startswith("-")), os.path.join(ROOT, "sage_amd", "csrc", "kernels.hip"))
    extra = [a for a in args if a.startswith("-")]
    lines = asm_of(src, extra)
    for k in kernels:
        print(f"# {k}")
        print(f"{'depth':>5} {'instr':>6} {'valu':>6} {'readlane':>9} {'writelane':>9} {'scratch_ld':>10} {'scratch_st':>10}")
        for d, r in sorted(analyse(lines, k).items()):
            print(f"{d:>5} {r['instr']:>6} {r['valu']:>6} {r['sgpr_reload']:>9} {r['sgpr_spill']:>9} {r['vgpr_reload']:>10} {r['vgpr_spill']:>10}")
