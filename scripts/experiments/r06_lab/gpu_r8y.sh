#!/bin/bash
# r8y: the peptide-major fragment list released after the index build and made again on demand (stream variant): the whole GPU suite; C3's index footprint and build time
OUT=gpurun_out/r8y; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 600 python - > $OUT/footprint.txt 2>&1 <<'PY'
import time, os, sys
sys.path.insert(0, os.getcwd())
import bench
from sage_amd.api import DeviceDatabase, Scorer
from sage_amd.workloads import CONFIGS, build_host_db, scorer_params
cfg = CONFIGS["C3"]
host = build_host_db(cfg, peptides_only=True)
for keep in ("1", None):
    if keep: os.environ["SAGE_HIP_KEEP_PM_FRAG"] = keep
    else: os.environ.pop("SAGE_HIP_KEEP_PM_FRAG", None)
    t0 = time.time(); dev = DeviceDatabase(host, 0, build_on_device=True); t1 = time.time()
    print("KEEP_PM_FRAG", keep, "index bytes on device %.3f GB" % (dev.device_bytes / 1e9), "build %.2f s" % (t1 - t0))
    if not keep:
        batch, _ = bench.generate_workload(cfg, host, 20000)
        os.environ["SAGE_HIP_NARROW"] = "stream"
        sc = Scorer(dev, scorer_params(cfg)); db = sc.upload(batch)
        t0 = time.time(); sc.score_resident(db); t1 = time.time()
        from sage_amd import _lib as L
        print("first stream-variant step (makes the list again) %.3f s; index bytes now %.3f GB" % (t1 - t0, int(L.load().sage_hip_db_device_bytes(dev._h)) / 1e9))
        t0 = time.time(); sc.score_resident(db); t1 = time.time()
        print("second step %.4f s" % (t1 - t0))
    dev.close()
PY
cat $OUT/footprint.txt | tail -6
