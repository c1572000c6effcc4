#!/bin/bash
# Same-box A/B of the default bench between the shipped library and a VARIANT build of it (python -m holoscene_amd.csrc.build --variant NAME -DFLAG...):
# alternating runs, mean / median ms per step and rays/s.     bash tools/ab_lib.sh NAME [runs] [extra bench args]
V=$1; RUNS=${2:-3}; shift 2 2>/dev/null
ARGS="--no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 200 --warmup 20 --roofline-steps 0 $@"
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq $RUNS); do
  for lib in "" holoscene_amd/csrc/libholoscene_hip_$V.so; do
    HOLOSCENE_LIB=${lib:+$PWD/$lib} python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-shipped}'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_median'), round(d['value']))"
  done
done
