#!/bin/bash
# build kernel variants: scripts/variants.sh name1:"-DX=0 -DY=1" name2:...   -> sage_amd/libsage_hip_<name>.so
# (knobs: SAGE_PRELIM_WAVES, SAGE_RESCORE_WAVES, SAGE_PROBE_PER_LANE, SAGE_TILE8_CELLS in kernels.hip; A/B them on the GPU with scripts/ab_libs.sh)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  SAGE_HIP_LIB=$PWD/sage_amd/libsage_hip_$name.so SAGE_HIP_OBJ_SUFFIX=_$name SAGE_HIP_EXTRA_FLAGS="$flags" python -m sage_amd.build --force 2>&1 | grep -i "error" &
done
wait; ls -la sage_amd/libsage_hip_*.so
