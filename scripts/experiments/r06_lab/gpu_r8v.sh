#!/bin/bash
# r8v: what IS the atomics' arrival order on C4?  (scripts/experiments/r06_lab/queue_order_probe.py)
OUT=gpurun_out/r8v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/experiments/r06_lab/queue_order_probe.py C4 100000 > $OUT/queue_order_C4.txt 2>&1; tail -12 $OUT/queue_order_C4.txt
