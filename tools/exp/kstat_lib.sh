#!/bin/bash
# per-kernel average durations of the default bench's hot loop, shipped library vs a variant build (rocprofv3 --kernel-trace --stats)
#   bash tools/exp/kstat_lib.sh VARIANT [kernel-name regex]
V=$1; PAT=${2:-.}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "" $R/holoscene_amd/csrc/libholoscene_hip_$V.so; do
  rm -rf /tmp/ksl
  HOLOSCENE_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksl -o k -- python $R/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 100 --warmup 10 --roofline-steps 0 > /tmp/ksl.log 2>&1
  echo "== ${lib:-shipped}"
  python - "$PAT" <<'PY'
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob("/tmp/ksl/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:60]
    if re.search(sys.argv[1], n): print(f'{int(r["Calls"]):7d} calls  avg {float(r["AverageNs"]) / 1000:8.2f} us  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms  {n}')
PY
done
