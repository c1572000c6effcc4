#!/bin/bash
# kernel sequence of a background-patch iteration (every 10th) of the default bench under rocprofv3
TAG=${1:-bg}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 60 --warmup 20 --roofline-steps 0 > $OUT/bench.log 2>&1 || true
TR=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
for k in 50 51 52 53 54 55 56 57 58 59; do python $GRAFT_REPO_ROOT/tools/iter_kernels.py $TR $k > /tmp/it_$k.txt 2>/dev/null; echo "$k $(wc -l < /tmp/it_$k.txt) $(sed -n 2p /tmp/it_$k.txt)"; done
BEST=$(for k in 50 51 52 53 54 55 56 57 58 59; do echo "$(wc -l < /tmp/it_$k.txt) $k"; done | sort -n | tail -1 | cut -d' ' -f2)
cp /tmp/it_$BEST.txt $OUT/bg_iteration_kernels.txt; cat $OUT/bg_iteration_kernels.txt
