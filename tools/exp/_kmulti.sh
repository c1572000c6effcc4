#!/bin/bash
# per-kernel averages (regex) of the bench loop for the shipped library and several variants:  bash tools/exp/_kmulti.sh REGEX v1 v2 ...
PAT=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" "$@"; do
  lib=${v:+$R/holoscene_amd/csrc/libholoscene_hip_$v.so}
  rm -rf /tmp/ksl
  HOLOSCENE_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksl -o k -- python $R/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 60 --warmup 10 --roofline-steps 0 > /tmp/ksl.log 2>&1
  python - "$PAT" "${v:-shipped}" <<'PY'
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob("/tmp/ksl/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:40]
    if re.search(sys.argv[1], n): print(f'{sys.argv[2]:8s} {int(r["Calls"]):6d} calls  avg {float(r["AverageNs"]) / 1000:8.2f} us  {n}')
PY
done
