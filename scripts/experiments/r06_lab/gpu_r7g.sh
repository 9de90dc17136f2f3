#!/bin/bash
# r7g: what the wavefront-per-query replay does on C5 / C4 (stream words, offers, cycles per query); the edge-case test with zero / tiny peak masses
OUT=gpurun_out/r7g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "edge_cases" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout 600 python scripts/replay_probe.py C5 40000 > $OUT/replay_C5.txt 2>&1; tail -3 $OUT/replay_C5.txt
timeout 600 python scripts/replay_probe.py C4 20000 > $OUT/replay_C4.txt 2>&1; tail -3 $OUT/replay_C4.txt
