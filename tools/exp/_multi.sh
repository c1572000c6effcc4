#!/bin/bash
# several variant libraries against the shipped one, one bench run each per round (scratch driver for tools/ab_lib.sh-style comparisons)
ARGS="--no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 200 --warmup 20 --roofline-steps 0"
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  for v in "" "$@"; do
    lib=${v:+$PWD/holoscene_amd/csrc/libholoscene_hip_$v.so}
    HOLOSCENE_LIB=$lib python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${v:-shipped}', d['ms_per_step'], d.get('ms_per_step_median'), round(d['value']))"
  done
done
