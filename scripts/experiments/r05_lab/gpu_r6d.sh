#!/bin/bash
# compiler scheduling / allocation flags on kernels.hip (scripts/variants.sh), C3: does another allocation of the knife's-edge kernels run faster?
OUT=gpurun_out/r6d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python scripts/ab_multi.py C3 --sizes c0/8,500000 --steps 40 -- base:AB_TIMING_EVERY=4 trackers:AB_TIMING_EVERY=4 noalign:AB_TIMING_EVERY=4 maxilp:AB_TIMING_EVERY=4 bias100:AB_TIMING_EVERY=4 prealloc:AB_TIMING_EVERY=4 ifcvt:AB_TIMING_EVERY=4 nohirp:AB_TIMING_EVERY=4 wprio:AB_TIMING_EVERY=4 base:AB_TIMING_EVERY=4 > $OUT/c3_flags.txt 2>&1; grep -E "^==|RESULT" $OUT/c3_flags.txt
