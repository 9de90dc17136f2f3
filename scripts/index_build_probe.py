"""Ad-hoc GPU probe: host vs device index construction time.  usage: python scripts/index_build_probe.py [config]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sage_amd.api import *
from sage_amd.synthetic import *
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
fasta = synthetic_fasta(cfg["proteins"], cfg["fasta_seed"])
t = time.time(); host = DatabaseParameters(**cfg["db"]).build(fasta); t_full = time.time() - t
t = time.time(); pep = DatabaseParameters(**cfg["db"]).build(fasta, peptides_only=True); t_pep = time.time() - t
t = time.time(); d1 = DeviceDatabase(host, 0); t_d1 = time.time() - t
d1.close()
t = time.time(); d2 = DeviceDatabase(pep, 0); t_d2 = time.time() - t
t = time.time(); d3 = DeviceDatabase(pep, 0); t_d3 = time.time() - t
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'C3'}: peptides {host.n_peptides} fragments {host.n_fragments}")
print(f"  host Parameters::build (digest..fragments, sort, buckets): {t_full:.2f} s ; peptides only: {t_pep:.2f} s")
print(f"  sage_hip_db_create from host fragments: {t_d1:.2f} s ; from peptides (device build): {t_d2:.2f} s (second time {t_d3:.2f} s)")
print(f"  end to end: host path {t_full + t_d1:.2f} s  ->  device path {t_pep + t_d3:.2f} s")
