#!/bin/bash
# A/B of an environment switch inside ONE gpurun call (box-to-box variance is 8-13 %): alternates the two values, prints the headline
# and the per-kernel averages of bench.py for each run.   usage: ab_env_bench.sh VAR value_a value_b [kernel-name-substring]
VAR=$1; A=$2; B=$3; K=${4:-sampler}
for v in $A $B $A $B; do
  env $VAR=$v python bench.py --steps 100 --warmup 15 --no-cpu-baseline --no-fp32-point 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ks=[(k['kernel'][:28],k['avg_us']) for k in d['roofline']['kernels'] if '$K' in k['kernel']]
print('$VAR=$v', d['ms_per_step'], d['ms_per_step_median'], ks)
"
done
