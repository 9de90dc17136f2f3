#!/bin/bash
# A campaign of randomised scorer configurations (tests/test_gpu_fuzz.py) beyond the suite's fixed cases: CASES per world under each
# SALT, HIP path against the oracle field for field.  Run through gpurun from the repo root; the log goes to gpurun_out/fuzz/.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_fuzz_campaign.sh 400 1 2 3'
CASES=${1:-400}; shift
SALTS=${@:-1}
mkdir -p gpurun_out/fuzz
for s in $SALTS; do
    SAGE_FUZZ_CASES=$CASES SAGE_FUZZ_SALT=$s timeout 1400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu --no-header -p no:cacheprovider \
        > gpurun_out/fuzz/salt_$s.log 2>&1
    echo "salt $s, $CASES cases per world: $(tail -1 gpurun_out/fuzz/salt_$s.log)"
    grep -E "^(FAILED|ERROR)" gpurun_out/fuzz/salt_$s.log | cut -c1-400 | head -20
done
