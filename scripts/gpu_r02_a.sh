#!/bin/bash
OUT=gpurun_out/r02a; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -1 $OUT/build.log
nproc; free -g | head -2
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -60 $OUT/pytest.log
scripts/gpu_calib.sh r02_calib
