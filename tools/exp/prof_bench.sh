#!/bin/bash
# tools/exp/prof_bench.sh PATTERN -- the default bench's hot loop under rocprofv3 --kernel-trace --stats; prints the kernels matching PATTERN
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o pb -- python $R/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --steps 100 --warmup 20 > /tmp/pb.log 2>&1
python - "$1" <<'PY'
import csv, glob, sys, re
f = glob.glob("/tmp/pb/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    tot += float(r["TotalDurationNs"])
    if re.search(sys.argv[1], r["Name"]):
        print(re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:50], r["Calls"], round(float(r["AverageNs"]) / 1000, 1), "us  min", round(float(r["MinNs"]) / 1000, 1))
PY
python $R/tools/iter_kernels.py $(find /tmp/pb -name "*kernel_trace.csv" | head -1) 60 2>/dev/null | grep -E "$1|span"
tail -1 /tmp/pb.log | cut -c100-230
