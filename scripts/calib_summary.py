#!/usr/bin/env python3
"""Merge scripts/calib_traffic's CALIB lines with the rocprofv3 --pmc passes taken over it -> markdown table on stdout.
usage: python scripts/calib_summary.py <calib_stdout.log> <pmc_dir> [<pmc_dir> ...]"""
import glob
import re
import sqlite3
import sys

rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"CALIB (\S+) requested_bytes=(\d+) sectors64_bytes=(\d+) ms=([\d.]+)", line)
    if m:
        rows[m.group(1)] = dict(req=float(m.group(2)), sec=float(m.group(3)), ms=float(m.group(4)))
ctrs = {}
for d in sys.argv[2:]:
    for db in glob.glob(d + "/**/*.db", recursive=True):
        con = sqlite3.connect(db)
        try:
            q = con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name")
            for name, ctr, avg, n in q:
                short = re.sub(r"^void ", "", name).split("(")[0].split("<")[0]
                ctrs.setdefault(short, {})[ctr] = avg
        except sqlite3.Error as e:
            print(f"(could not read {db}: {e})")
names = sorted({c for v in ctrs.values() for c in v})
print("| kernel | requested MB | touched 64 B sectors MB | ms | GB/s (sectors) | " + " | ".join(names) + " | FETCH_SIZE·1024 / sectors | WRITE_SIZE·1024 / sectors |")
print("|---|---|---|---|---|" + "---|" * (len(names) + 2))
for k, r in rows.items():
    key = {"gather_read4": "gather_read", "gather_read8": "gather_read", "gather_read16": "gather_read"}.get(k, k)
    c = ctrs.get(k) or ctrs.get(key) or {}
    f = c.get("FETCH_SIZE")
    w = c.get("WRITE_SIZE")
    print(f"| {k} | {r['req']/1e6:.1f} | {r['sec']/1e6:.1f} | {r['ms']:.3f} | {r['sec']/r['ms']/1e6:.0f} | " +
          " | ".join(f"{c.get(n, float('nan')):.4g}" for n in names) +
          f" | {f*1024/r['sec']:.3f} |" if f else " | n/a |", f" {w*1024/r['sec']:.3f} |" if w else " n/a |")
print("\n(the three gather_read<T> instantiations share one kernel name in the counter tables when rocprofv3 strips template "
      "arguments; see the per-kernel dump below)\n")
for k, v in sorted(ctrs.items()):
    print(f"- `{k}`: " + ", ".join(f"{n}={x:.6g}" for n, x in sorted(v.items())))
