#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output of a `--kernel-trace --stats` run of bench.py into the small text summary that is
committed under profiles/, next to the figures of the bench line of the same configuration (HIP-event kernel times, live PMC
traffic), so that the two can be held against each other.
Usage: python profiles/summarize_rocprof.py <tag> <config> <trace.db> [<bench.json>]"""
import json
import sqlite3
import sys


def short(name):
    return name.replace("sagehip::(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:70]


def main():
    tag, config, trace = sys.argv[1], sys.argv[2], sys.argv[3]
    bench = sys.argv[4] if len(sys.argv) > 4 else None
    out = [f"# rocprofv3 --kernel-trace --stats -- python bench.py --config {config} --steps 5 --warmup 2 --no-extras   ({tag})\n",
           "# 7 calls of score_resident (2 warm-up + 5 timed); narrow_kernel is the exact retry pass, the large-window kernels show twice per call\n"]
    con = sqlite3.connect(trace)
    out.append(f"{'kernel':<70} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'pct':>6}\n")
    per_step = {}
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        s = short(name)
        out.append(f"{s:<70} {calls:>6} {total / 1e3:>10.2f} {avg:>10.1f} {pct:>6.2f}\n")   # the view is in microseconds
        if s.startswith(("prelim_", "tile_", "rescore", "schedule", "narrow_", "search_")):
            per_step[s] = total / 1e3 / 7.0
    out.append("\n# search kernels, ms per step (total / 7 calls):\n")
    for k, v in sorted(per_step.items(), key=lambda kv: -kv[1]):
        out.append(f"  {k:<40} {v:8.3f}\n")
    try:
        # Launch geometry only.  rocprofv3's scratch_size / vgpr_count / sgpr_count columns are allocation granules of the dispatch
        # packet (round 3's summaries printed `scratch=1024 vgpr=48 sgpr=112` for every narrow kernel), not the kernel's registers
        # and spills: those are in the compiler's own table, profiles/<tag>_kernel_resources.txt (scripts/kernel_resources.py).
        out.append("\n# per-dispatch launch geometry (registers / spills / scratch: the compiler's table, profiles/%s_kernel_resources.txt)\n" % tag)
        q = ("select name, grid_x, workgroup_x, lds_size, count(*) from kernels group by name, grid_x, workgroup_x, lds_size")
        for r in con.execute(q):
            if short(r[0]).startswith(("prelim_", "tile_", "rescore", "narrow_", "search_")):
                out.append(f"  {short(r[0]):<40} grid={r[1]} wg={r[2]} dynamic_lds={r[3]} n={r[4]}\n")
    except sqlite3.Error as e:
        out.append(f"(geometry query failed: {e})\n")
    if bench:
        try:
            j = json.loads([ln for ln in open(bench) if ln.startswith("{")][-1])
            rf = j["roofline"]
            out.append("\n# the bench line of the same build (profiles/%s_%s_bench.json): HIP events on the scorer's stream\n" % (tag, config))
            out.append(f"  value {j['value']:.4g} {j['unit']}, ms_per_step {j['ms_per_step']:.3f}, kernel_ms {rf['kernel_ms']}\n")
            out.append(f"  algorithmic bytes/spectrum {rf['algorithmic_bytes_per_spectrum']}\n")
            out.append(f"  traffic bytes/spectrum (PMC) {rf.get('traffic_bytes_per_spectrum')}  [{rf.get('traffic_source')}]\n")
            out.append(f"  GPU-algorithm bytes/spectrum {rf.get('gpu_algorithm_bytes_per_spectrum')}\n")
            out.append(f"  frac (algorithmic) {rf['frac']:.3f}  frac_traffic {rf.get('frac_traffic')}  frac_gpu_algorithm {rf.get('frac_gpu_algorithm')}\n")
        except (OSError, ValueError, KeyError, IndexError) as e:
            out.append(f"(bench line not read: {e!r})\n")
    print("".join(out))


if __name__ == "__main__":
    main()
