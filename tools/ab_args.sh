#!/bin/bash
# Same-box A/B of the default bench between two ARGUMENT settings: alternating runs.   bash tools/ab_args.sh "<args A>" "<args B>" [runs] [extra bench args]
A=$1; B=$2; RUNS=${3:-3}; shift 3 2>/dev/null
ARGS="--no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 200 --warmup 20 --roofline-steps 0 $@"
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq $RUNS); do
  for v in "$A" "$B"; do
    python bench.py $ARGS $v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$v]', d['ms_per_step'], d.get('ms_per_step_median'), round(d['value']), 'regular', d['iteration_kinds']['regular']['ms_mean'], 'bg', d['iteration_kinds']['background_patch']['ms_mean'])"
  done
done
