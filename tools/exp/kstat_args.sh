#!/bin/bash
# per-kernel totals of the bench's hot loop at other arguments (rocprofv3 --kernel-trace --stats):   bash tools/exp/kstat_args.sh <bench args...>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ka
STEPS=100
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka -o k -- python $R/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps $STEPS --warmup 10 --roofline-steps 0 "$@" > /tmp/ka.log 2>&1
python - $STEPS <<'PY'
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob("/tmp/ka/**/*kernel_stats.csv", recursive=True)[0])))
it = int(sys.argv[1]) + 10 + 6      # timed + warm-up + capture passes (approximate divisor)
for r in rows[:40]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:70]
    print(f'{float(r["TotalDurationNs"]) / it / 1000:8.1f} us/iter  x{int(r["Calls"]) / it:5.1f}  avg {float(r["AverageNs"]) / 1000:8.2f} us  {n}')
PY
tail -1 /tmp/ka.log | cut -c1-200
