"""End-to-end run of the JSON-config CLI on a synthetic, C2-sized data set (BASELINE.json configs[1]: yeast-like digest,
50 000 MS2 spectra in two mzML files, one of them gzip-compressed): FASTA + mzML on disk in, results.sage.tsv out, with the
stage times the CLI logs (reader, preprocessing, search, rescoring, writers).  usage: python scripts/cli_e2e.py [outdir] [n]"""
import gzip, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sage_amd.mzml import write_mzml
from sage_amd.synthetic import synthetic_fasta, synthetic_spectra
from sage_amd.workloads import CONFIGS, build_host_db

out = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="sage_e2e_", dir="/tmp")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
cfg = CONFIGS["C2"]
os.makedirs(out, exist_ok=True)
t0 = time.time()
fasta = synthetic_fasta(cfg["proteins"], cfg["fasta_seed"])
open(os.path.join(out, "db.fasta"), "w").write(fasta)
host = build_host_db(cfg, peptides_only=True)
paths = []
for k in range(2):
    p = os.path.join(out, f"run{k}.mzML")
    write_mzml(p, synthetic_spectra(host, n // 2, cfg["spectra_seed"] + k))
    if k == 1:
        with open(p, "rb") as fi, gzip.open(p + ".gz", "wb", compresslevel=1) as fo:
            shutil.copyfileobj(fi, fo)
        os.remove(p)
        p += ".gz"
    paths.append(p)
print(f"generated the data set in {time.time() - t0:.1f} s: " + ", ".join(f"{os.path.basename(p)} {os.path.getsize(p) >> 20} MiB" for p in paths), flush=True)
conf = {"database": {"bucket_size": 8192, "enzyme": {"missed_cleavages": 1, "min_len": 5, "max_len": 50, "cleave_at": "KR", "restrict": "P"},
                     "peptide_min_mass": 500.0, "peptide_max_mass": 5000.0, "static_mods": {"C": 57.0215}, "generate_decoys": True,
                     "fasta": os.path.join(out, "db.fasta")},
        "precursor_tol": {"ppm": [-10, 10]}, "fragment_tol": {"ppm": [-10, 10]}, "report_psms": 1, "mzml_paths": paths,
        "output_directory": os.path.join(out, "results")}
json.dump(conf, open(os.path.join(out, "config.json"), "w"))
for rep in ("cold", "warm"):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "sage_amd.cli", os.path.join(out, "config.json")], cwd=ROOT, capture_output=True, text=True)
    wall = time.time() - t0
    print(f"---- {rep} run: wall {wall:.2f} s, rc {r.returncode}")
    print(r.stdout[-3000:] if r.returncode == 0 else r.stderr[-3000:])
