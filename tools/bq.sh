#!/bin/bash
# quick bench: prints ms/step + kernel split; args passed to bench.py
python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernels_ms_per_step'].items()}, round(d['roofline']['frac'],3), d.get('half_iteration_kernel_ms'))
except Exception as e: print('FAILED', t[-600:])
"
