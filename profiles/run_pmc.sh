#!/bin/bash
# profiles/run_pmc.sh <tag> -- PMC passes (separate runs, --kernel-trace only) for HBM traffic of the hot kernels.
# FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 (TCC slots), so two passes.
set -e
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$C -o $C -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-graph --steps 3 --warmup 2 > $OUT/bench_$C.log 2>&1 || true
  python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/pmc_${TAG}_$C/*counter_collection.csv')
agg = collections.defaultdict(lambda: [0, 0.0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if r.get('Counter_Name') != '$C': continue
        k = r['Kernel_Name'][:70] + ' grid=' + r.get('Grid_Size', r.get('Grid_Size_X', '?'))
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
with open('$OUT/$C.csv', 'w') as o:
    o.write('kernel,dispatches,total_$C,avg_$C\n')
    for k, (n, v) in sorted(agg.items(), key=lambda t: -t[1][1])[:40]:
        o.write(f'"{k}",{n},{v},{v/n}\n')
print(open('$OUT/$C.csv').read()[:1500])
PY
done
