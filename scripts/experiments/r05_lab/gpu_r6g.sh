#!/bin/bash
# the lanes' own hits walked item by item (set charges only; no loop over the kinds for b / y) instead of ion by ion x every charge: C3 (run twice: two-form and one-form patch)
OUT=gpurun_out/r6g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_multi.py C3 --sizes c0/8,500000 --steps 40 -- base:AB_TIMING_EVERY=4 items:AB_TIMING_EVERY=4 base:AB_TIMING_EVERY=4 items:AB_TIMING_EVERY=4 > $OUT/c3_items.txt 2>&1; grep -E "^==|RESULT" $OUT/c3_items.txt
