#!/bin/bash
# profiles/run_rocprof.sh <tag> -- run on the GPU box (via gpurun): kernel-trace + stats of bench.py,
# keeps only the per-kernel summary (the raw trace is too large to travel back).
set -e
TAG=${1:-r}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point "$@" > $OUT/bench.log 2>&1 || true
find /tmp/prof_$TAG -name "*stats*.csv" -exec cp {} $OUT/ \;
ls -la /tmp/prof_$TAG/* | head; tail -2 $OUT/bench.log
