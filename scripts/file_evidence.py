"""File a GPU evidence pass (scripts/gpu_r6_evidence.sh -> gpurun_out/<tag>/) under profiles/ and bring the documents that quote it up to
date: profiles/<tag>_*, the compiler's resource table, the table of profiles/README.md (profiles/make_table.py) and the M spectra/s
figures of README.md's round summary (tests/test_docs_refs.py holds those to the committed bench lines).
usage: python scripts/file_evidence.py r06"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", tag)
for f in glob.glob(os.path.join(src, tag + "_*")):
    shutil.copy(f, os.path.join(ROOT, "profiles", os.path.basename(f)))
res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kernel_resources.py")], capture_output=True, text=True).stdout
open(os.path.join(ROOT, "profiles", tag + "_kernel_resources.txt"), "w").write(res)


def line(cfg):
    return json.loads([l for l in open(os.path.join(ROOT, "profiles", f"{tag}_{cfg}_bench.json")) if l.startswith("{")][-1])


vals = {c: line(c) for c in ("C3", "C3T", "C2", "C4", "C5")}
readme = os.path.join(ROOT, "README.md")
s = open(readme, encoding="utf-8").read()
i = s.index("* Round 6 on one MI355X")
j = s.index("\n* ", i + 1) if "\n* " in s[i + 1:] else len(s)
para = s[i:j]


def put(pattern, value, digits):
    global para
    m = re.search(pattern, para, flags=re.S)
    assert m, pattern
    para = para[:m.start(1)] + f"{value:.{digits}f}" + para[m.end(1):]


put(r"±10 ppm\) \*\*([\d.]+) M spectra/s\*\* resident", vals["C3"]["value"] / 1e6, 1)
h2h = vals["C3"]["host_to_host_value"]
put(r"\*\*([\d.]+) M host to host\*\*", (h2h["page_locked"] if isinstance(h2h, dict) else h2h) / 1e6, 1)
put(r"\(C3T\) ([\d.]+) M", vals["C3T"]["value"] / 1e6, 1)
put(r"C2 ([\d.]+) M", vals["C2"]["value"] / 1e6, 1)
put(r"open search C4 ([\d.]+) M", vals["C4"]["value"] / 1e6, 2)
put(r"wide-window C5 ([\d.]+) M", vals["C5"]["value"] / 1e6, 2)
ratio = vals["C3"]["value"] / vals["C3"]["cpu_baseline"]["value"]
m = re.search(r"([\d ]+)× the\s+restated CPU path", para)
assert m
para = para[:m.start(1)] + f"{ratio:,.0f}".replace(",", " ") + para[m.end(1):]
open(readme, "w", encoding="utf-8").write(s[:i] + para + s[j:])
pr = os.path.join(ROOT, "profiles", "README.md")
t = open(pr, encoding="utf-8").read()
tab = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "make_table.py"), tag], capture_output=True, text=True).stdout.strip()
a = t.index("| config | spectra (1 GPU)")
b = t.index("\n\n", a)
open(pr, "w", encoding="utf-8").write(t[:a] + tab + t[b:])
for c, v in vals.items():
    print(c, f"{v['value'] / 1e6:.2f} M spectra/s", f"{v['ms_per_step']:.3f} ms/step")
