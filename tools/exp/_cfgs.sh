#!/bin/bash
# other BASELINE configs' shapes on one GPU (points of DESIGN's table; not bench lines)
F="--no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --roofline-steps 0"
run() { echo "== $*"; python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['iteration_kinds'])"; }
run --rays 512
run --rays 1024 --objects 21
run --rays 4096 --steps 100
run --rays 2048 --samples 192 --steps 100
run --rays 2048 --samples 192 --eikonal fd --precision fp32 --steps 30
run --rays 1024 --objects 64
