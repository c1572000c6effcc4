#!/bin/bash
# per-kernel time of the fp32 configuration's iteration (rocprofv3 --kernel-trace --stats), top 30 by share
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o pf -- python $R/bench.py --precision fp32 --no-cpu-baseline --no-second-point --steps 30 --warmup 6 > /tmp/pf.log 2>&1
python - <<'PY'
import csv, glob, re
rows = list(csv.DictReader(open(glob.glob("/tmp/pf/**/*kernel_stats.csv", recursive=True)[0])))
it = 36 + 6       # timed + warm-up iterations (approximately: everything the process launched)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms per iteration ~", round(tot / it / 1e6, 2))
for r in rows[:32]:
    print(f'{float(r["TotalDurationNs"]) / it / 1000:8.1f} us/iter  x{int(r["Calls"]) / it:6.1f}  avg {float(r["AverageNs"]) / 1000:8.1f}  ' + re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Name"])[:100])
PY
tail -1 /tmp/pf.log | cut -c100-220
