#!/usr/bin/env python3
"""Print the markdown table of profiles/README.md from the committed bench lines: python profiles/make_table.py r02"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
here = os.path.dirname(os.path.abspath(__file__))
print("| config | spectra (1 GPU) | spectra/s resident | sustained | ms/step | prelim / rescore ms | bytes/spectrum: reference algorithm (§8d) / "
      "asked for by the kernels / moved from HBM (PMC) | fraction of 8 TB/s, same three counts (prelim phase) | host to host, page-locked | "
      "CPU port: best (threads), 1 thread | tied spectra re-run exactly |" + (" issue slots of the dominant kernel; whole-workload parity |" if tag >= "r05" else ""))
print("|---|---|---|---|---|---|---|---|---|---|---|" + ("---|" if tag >= "r05" else ""))
for c in ("C3", "C2", "C3T", "C4", "C5"):
    path = os.path.join(here, f"{tag}_{c}_bench.json")
    if not os.path.exists(path):
        continue
    j = json.loads([ln for ln in open(path) if ln.startswith("{")][-1])
    rf, cpu = j["roofline"], j["cpu_baseline"]
    p = rf["by_kernel"]["prelim"]
    ab, tb, gb = rf["algorithmic_bytes_per_spectrum"], rf["traffic_bytes_per_spectrum"] or {}, rf["gpu_algorithm_bytes_per_spectrum"] or {}
    kb = lambda v: "—" if v is None else (f"{v / 1e6:.2f} MB" if v >= 1e6 else f"{v / 1e3:.1f} KB")
    fr = lambda v: "—" if v is None else f"{v:.2f}"
    n = rf["routing"]["spectra"]
    print(f"| {c} | {n:,} | **{j['value'] / 1e6:.2f} M** | {j['sustained']['value'] / 1e6:.2f} M | {j['ms_per_step']:.2f} | "
          f"{rf['kernel_ms']['prelim']:.2f} / {rf['kernel_ms']['rescore']:.2f} | "
          f"{kb(ab['prelim'])} / {kb(gb.get('prelim'))} / {kb(tb.get('prelim'))} | "
          f"{fr(p['frac'])} / {fr(p['frac_gpu_algorithm'])} / {fr(p['frac_traffic'])} | "
          f"{j['host_to_host_value']['page_locked'] / 1e6:.2f} M | "
          f"{cpu['value'] / 1e3:.1f} k ({cpu['cores']}), {cpu['threads_table']['1']['spectra_per_s'] / 1e3:.2f} k | "
          f"{100.0 * rf['routing']['exact_retry_for_tied_hyperscores'] / n:.1f} % |" +
          (f" issue slots {rf['frac_issue_slots']:.2f} ({rf['issue']['kernel'].split('<')[0]}); parity {j['parity']['spectra_checked']:,} spectra / "
           f"{j['parity']['psms']:,} PSMs |" if rf.get("frac_issue_slots") is not None and j.get("parity") else ""))
