#!/bin/bash
# What the GPU box's host offers a process: CPU count, affinity, cgroup quota, NUMA layout (for bench.py's cpu_baseline)
python - <<'PY'
import sys; sys.path.insert(0, '.')
import bench; print("host_cpu_budget:", bench.host_cpu_budget())
PY
nproc; lscpu | grep -E "^(CPU\(s\)|Thread|Core|Socket|NUMA|Model name)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /proc/self/status | grep -E "Cpus_allowed_list|Mems_allowed_list"
