#!/bin/bash
# A/B of kernel build variants through scripts/ab_env.py: scripts/ab_libs_env.sh <config> <n_spectra|0> name1 name2 ...  ("base" = libsage_hip.so)
CFG=${1:-C3}; N=${2:-0}; shift 2
[ "$N" = 0 ] && N=""
for name in "$@"; do
  lib=$PWD/sage_amd/libsage_hip_$name.so; [ "$name" = base ] && lib=$PWD/sage_amd/libsage_hip.so
  echo -n "$name: "; SAGE_HIP_LIB=$lib timeout 300 python scripts/ab_env.py $CFG $N 2>&1 | tail -1
done
