#!/bin/bash
# Same-box A/B of an environment switch on this tree: bash tools/ab_env_bench2.sh "VAR=a" "VAR=b" [runs]
A=$1; B=$2; RUNS=${3:-3}
ARGS="--no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 200 --warmup 20 --roofline-steps 0"
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq $RUNS); do
  for e in "$A" "$B"; do
    env $e python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['ms_per_step'], d.get('ms_per_step_median'), round(d['value']))"
  done
done
