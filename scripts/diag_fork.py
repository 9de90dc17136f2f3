import sys, os, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
from sage_amd.api import DatabaseParameters, DeviceDatabase
from sage_amd.synthetic import synthetic_fasta
host = DatabaseParameters(static_mods={"C": 57.0215}).build(synthetic_fasta(50, seed=3), peptides_only=True)
def f(x): return x * 2
if "fork" in mode:
    with mp.get_context("fork").Pool(4) as pool: print(pool.map(f, range(4)))
if "torchfirst" not in mode and "notorch" not in mode:
    import torch; print("torch avail", torch.cuda.is_available()); torch.cuda.set_device(0)
try:
    dev = DeviceDatabase(host, 0, build_on_device=True); print(mode, "OK", dev.device_bytes)
except Exception as e:
    print(mode, "FAILED", e)
