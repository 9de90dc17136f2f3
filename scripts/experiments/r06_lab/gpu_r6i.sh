#!/bin/bash
# r6i: HBM traffic (L2 misses: 2 x FETCH_SIZE + WRITE_SIZE, KiB) of prelim_kernel with the succinct position table against round 5's
# row-major one — 131 072 C3 spectra, one part per step.
export TMPDIR=/tmp SAGE_HIP_WAYS=1
CMD="python bench.py --config C3 --spectra 131072 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-extras"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== succinct $C"; PMC_TIMEOUT=400 scripts/prof_pmc.sh r6i_new_$C "$C" $CMD | grep "prelim_kernel\|rescore_kernel"
  echo "== round 5 $C"; SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_r5.so PMC_TIMEOUT=400 scripts/prof_pmc.sh r6i_old_$C "$C" $CMD | grep "prelim_kernel\|rescore_kernel"
done
