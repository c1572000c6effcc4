#!/bin/bash
# Same-box A/B of the default bench between this tree and a second checkout (default .ab_old = the previous commit, built in place):
# alternating runs, medians of the regular iteration.   bash tools/ab_bench.sh [other_dir] [runs] [extra bench args]
# The second checkout is not kept in the tree: git worktree add .ab_old <rev> && (cd .ab_old && python -c "import __graft_entry__ as g; g.build()")
# (.ab_old is git-ignored; it travels with gpurun, ~80 MB per call, so remove it again: git worktree remove --force .ab_old)
OTHER=${1:-.ab_old}; RUNS=${2:-3}; shift 2 2>/dev/null
ARGS="--no-cpu-baseline --no-second-point --no-fp32-point --steps 200 --warmup 20 --roofline-steps 0 $@"
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq $RUNS); do
  for d in $OTHER .; do
    (cd $d && python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', d['ms_per_step'], d.get('ms_per_step_median'), round(d['value']), 'regular', d['iteration_kinds']['regular']['ms_mean'], 'bg', d['iteration_kinds']['background_patch']['ms_mean'])")
  done
done
