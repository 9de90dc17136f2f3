#!/bin/bash
# r8t: cycles per offer of the wavefront replay now (scripts/replay_probe.py: C5 40 000 and C4 20 000 spectra)
OUT=gpurun_out/r8t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/replay_probe.py C5 40000 > $OUT/replay_C5.txt 2>&1; tail -1 $OUT/replay_C5.txt
timeout 600 python scripts/replay_probe.py C4 20000 > $OUT/replay_C4.txt 2>&1; tail -1 $OUT/replay_C4.txt
