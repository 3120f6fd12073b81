#!/bin/bash
# quick bench lines for several workloads/modes: tools/bq2.sh "<bench args>" ["<bench args>" ...]
for a in "$@"; do
python bench.py --no-cpu-baseline --steps 3 --warmup 1 $a 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print('$a', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernels_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'],3), 'dualrows', d.get('rows_dual_per_step'), 'refined', d.get('rows_refined_per_step'), 'eig', round(d.get('eigen_host_ms_per_step',0),2), d.get('half_iteration_kernel_ms'), d['reconstruction_error'].get('mean'), d['reconstruction_error'].get('planted_part'), 'nnz', d['config']['nnz'])
except Exception as e: print('FAILED', '$a', t[-600:])
"
done
