#!/bin/bash
# tools/exp/ab_wgrad.sh "ENV=..." "ENV=..." : alternate the settings twice on one box, k_wgrad_pairs medians by caller + the bench's median step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for setting in "$@"; do
  rm -rf /tmp/ab
  env $setting timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab -o ab -- python $R/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --steps 150 --warmup 20 > /tmp/ab.log 2>&1
  echo "$setting :: $(python $R/tools/exp/wgrad_split.py /tmp/ab)"
done; done
