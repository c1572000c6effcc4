#!/bin/bash
# tools/exp/ab_bench.sh PATTERN "ENV=..." ... : alternate the settings twice on one box; kernels matching PATTERN (rocprof stats) + median step
R=$GRAFT_REPO_ROOT
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for setting in "$@"; do
  rm -rf /tmp/ab
  env $setting timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab -o ab -- python $R/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --steps 150 --warmup 20 > /tmp/ab.log 2>&1
  echo "== $setting :: $(python $R/tools/exp/wgrad_split.py /tmp/ab) :: $(tail -1 /tmp/ab.log | grep -o '"ms_per_step_median": [0-9.]*')"
  python - "$PAT" <<'PY'
import csv, glob, sys, re
f = glob.glob("/tmp/ab/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[1], r["Name"]):
        print("   ", re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:28], r["Calls"], round(float(r["AverageNs"]) / 1000, 1))
PY
done; done
