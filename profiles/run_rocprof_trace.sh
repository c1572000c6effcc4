#!/bin/bash
# like run_rocprof.sh but also brings the raw kernel trace back (use few steps!)
set -e
TAG=${1:-t}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point "$@" > $OUT/bench.log 2>&1 || true
find /tmp/prof_$TAG -name "*.csv" -exec cp {} $OUT/ \;
tail -1 $OUT/bench.log
