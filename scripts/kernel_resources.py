"""The compiler's own per-kernel resource table (registers, spills, scratch, occupancy) for a HIP translation unit:

    python scripts/kernel_resources.py [sage_amd/csrc/kernels.hip] [-D...]  > profiles/rNN_kernel_resources.txt

hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the flags of sage_amd/build.py.  No GPU needed.  (rocprofv3's per-dispatch
`scratch / vgpr / sgpr` columns are allocation granules of the dispatch packet, not these numbers.)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-Rpass-analysis=kernel-resource-usage", *extra, "-c", src, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", err)[1:]:
        name = blk.split("\n")[0].split(" [")[0].strip()

        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else -1
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("sagehip::(anonymous namespace)::", "").replace("sagehip::", "").replace("void ", "")
        dem = re.sub(r"\((?!anonymous).*$", "", dem)
        rows.append((dem, g("SGPRs"), g("VGPRs"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
                     g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    src = next((a for a in args if not a.startswith("-")), os.path.join(ROOT, "sage_amd", "csrc", "kernels.hip"))
    extra = [a for a in args if a.startswith("-")]
    print(f"# hipcc -Rpass-analysis=kernel-resource-usage --offload-arch=gfx950 -O3 -ffp-contract=off {' '.join(extra)} {os.path.relpath(src, ROOT)}")
    print(f"{'kernel':<58} {'sgpr':>5} {'vgpr':>5} {'sgpr_spill':>10} {'vgpr_spill':>10} {'scratch_B':>9} {'waves/SIMD':>10} {'static_lds':>10}")
    for r in table(src, extra):
        print(f"{r[0][:58]:<58} {r[1]:>5} {r[2]:>5} {r[3]:>10} {r[4]:>10} {r[5]:>9} {r[6]:>10} {r[7]:>10}")
