#!/bin/bash
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" | tail -1
for m in plain fork notorch notorch_fork; do timeout 120 python scripts/diag_fork.py $m 2>&1 | grep -v amdgpu.ids | tail -3; done
ldd sage_amd/libsage_hip.so | grep -i hip
python -c "import torch,os; print(torch.__file__); print([f for f in os.listdir(os.path.join(os.path.dirname(torch.__file__),'lib')) if 'hip' in f][:10])"
