#!/bin/bash
# r6f: the default mass-block plan (sharding.plan_mass_shards, 8 ranks) on workloads it was NOT tuned on — C2 (50 000 spectra,
# yeast-like) and the tie-rich C3T — against C3: ms per step of each of the eight shards (VERDICT r05 weak 6 / task 5a).
OUT=gpurun_out/r6f; mkdir -p $OUT; export TMPDIR=/tmp
S=c0/8,c1/8,c2/8,c3/8,c4/8,c5/8,c6/8,c7/8
for C in C3T C2 C3; do
  AB_LIGHT_REFINE=8 timeout 1500 python scripts/ab_multi.py $C --sizes $S --steps 30 -- base > $OUT/shards_$C.log 2>&1; echo "$C rc=$?"
  AB_LIGHT_REFINE=1 timeout 1500 python scripts/ab_multi.py $C --sizes $S --steps 30 -- base > $OUT/shards_${C}_norefine.log 2>&1; echo "$C (no refine) rc=$?"
  grep "RESULT\|SLICE" $OUT/shards_$C.log | paste - - | awk '{print $2, $3, $4, $5, "|", $14, $15, $16, $17}' 
  echo "-- without the light-end refinement"
  grep "RESULT" $OUT/shards_${C}_norefine.log | awk '{print $2, $3, $4, $5, $6}'
done
