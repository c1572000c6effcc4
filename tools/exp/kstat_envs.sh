#!/bin/bash
# per-kernel average durations of the default bench's hot loop under SEVERAL settings of one environment switch:   bash tools/exp/kstat_envs.sh VAR 'regex' v1 v2 ...
VAR=$1; PAT=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/ks_$v
  env $VAR=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o k -- python $R/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 100 --warmup 10 --roofline-steps 0 > /tmp/ks_$v.log 2>&1
  echo "== $VAR=$v  $(tail -1 /tmp/ks_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'))" 2>/dev/null)"
  python - "$v" "$PAT" <<'PY'
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob(f"/tmp/ks_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:60]
    if re.search(sys.argv[2], n): print(f'{int(r["Calls"]):7d} calls  avg {float(r["AverageNs"]) / 1000:8.2f} us  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms  {n}')
PY
done
