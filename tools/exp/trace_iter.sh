#!/bin/bash
# one iteration's kernel sequence of the default bench under rocprofv3: bash tools/exp/trace_iter.sh <tag>
TAG=${1:-t}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 100 --warmup 20 --roofline-steps 0 > $OUT/bench.log 2>&1 || true
TR=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/iter_kernels.py $TR 60 > $OUT/iteration_kernels.txt 2>&1 || true
cat $OUT/iteration_kernels.txt
