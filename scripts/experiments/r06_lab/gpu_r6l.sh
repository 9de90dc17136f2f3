#!/bin/bash
# r6l: where C5's 56 ms go — kernel trace of two full-size steps (the retry pass's kernels are the second launch of each name)
export TMPDIR=/tmp
scripts/prof_cmd.sh r6l python bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-extras
