for f in 0 4; do echo "== flags $f"; SAGE_HIP_DEBUG_FLAGS=$f python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'] and d['roofline']['kernel_ms'] or d['ms_per_step'])"; done
SAGE_HIP_DEBUG_FLAGS=0 python bench.py --config C3 --steps 5 --warmup 2 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['cpu_baseline']['parity'])"
