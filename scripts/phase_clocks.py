import os, sys, numpy as np, ctypes as C
os.environ["SAGE_HIP_PHASE_CLOCKS"]="1"
sys.path.insert(0, os.getcwd())
from sage_amd import _lib as L
from sage_amd.api import *
from sage_amd.synthetic import *
import bench
cfg=bench.CONFIGS["C2"]
host=DatabaseParameters(**cfg["db"]).build(synthetic_fasta(cfg["proteins"], cfg["fasta_seed"]))
sp=SpectrumProcessor(150,True,0.0)
batch=SpectrumBatch.from_spectra([p for p in (sp.process(r) for r in synthetic_spectra(host, 20000, cfg["spectra_seed"])) if len(p.masses)>=15])
scorer=Scorer(DeviceDatabase(host,0), ScorerParams()); db=scorer.upload(batch)
for _ in range(3): scorer.score_resident(db)
out=np.zeros(32,np.uint64); L.check(L.load().sage_hip_debug_phase_cycles(scorer._h, L.as_ptr(out,C.c_uint64)))
n=3*min(batch.n,4096)
print("prelim  cycles/spectrum: staging %d search %d match %d trim %d output %d" % tuple(out[:5]//n))
print("rescore cycles/spectrum: setup %d phaseA %d phaseB %d rank %d emit %d" % tuple(out[8:13]//n))
print(scorer.last_timing())
