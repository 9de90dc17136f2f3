#!/bin/bash
# r8v: what IS the atomics' arrival order on C4?  (a probe script + a lab export of the queue, sage_hip_debug_queue, both removed again
# after the call: this file is the record of what was run, not runnable any more — its result is in RESULTS.md)
OUT=gpurun_out/r8v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/experiments/r06_lab/queue_order_probe.py C4 100000 > $OUT/queue_order_C4.txt 2>&1; tail -12 $OUT/queue_order_C4.txt
