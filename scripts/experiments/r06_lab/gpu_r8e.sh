#!/bin/bash
# r8e: the large-window kernels' queue built by a kernel of its own (a wavefront's spectra per atomic) instead of one returning atomic
# per spectrum on one address inside prelim_kernel; the whole GPU suite, C5 / C4 against the build before (q5)
OUT=gpurun_out/r8e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 1500 python scripts/ab_multi.py C5 --sizes 200000 --steps 5 -- q5 base > $OUT/ab_C5.log 2>&1; grep RESULT -B1 $OUT/ab_C5.log
timeout 1500 python scripts/ab_multi.py C4 --sizes 100000 --steps 5 -- q5 base > $OUT/ab_C4.log 2>&1; grep RESULT -B1 $OUT/ab_C4.log
