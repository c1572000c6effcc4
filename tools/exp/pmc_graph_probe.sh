#!/bin/bash
# does a --pmc pass see the kernels of the replayed whole-iteration graph?
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_probe -o F -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 12 --warmup 2 --roofline-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_probe.log 2>&1
echo rc=$?
F=$(find /tmp/pmc_probe -name "*counter_collection.csv" | head -1)
echo $F; wc -l $F
python - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][:40]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{k:42s} n={len(v):4d} avg={sum(v)/len(v):12.1f}")
PY
