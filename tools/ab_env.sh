#!/bin/bash
# Same-box A/B of the default bench between two settings of ONE environment switch: alternating runs.   bash tools/ab_env.sh VAR A B [runs] [extra bench args]
VAR=$1; A=$2; B=$3; RUNS=${4:-3}; shift 4 2>/dev/null
ARGS="--no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 200 --warmup 20 --roofline-steps 0 $@"
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq $RUNS); do
  for v in $A $B; do
    env $VAR=$v python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d.get('ms_per_step_median'), round(d['value']), 'bg', d['iteration_kinds']['background_patch']['ms_mean'])"
  done
done
