#!/bin/bash
# profiles/run_pmc_tcp.sh <tag> -- what bounds the hash gathers?  Vector-L1 (TCP) and texture-addresser (TA) counters of the hand-written hash
# kernels, each counter set in its own pass with --kernel-trace only (as MI355X_MICROARCH.md prescribes), over the eager iteration of the
# default bench.  TCP_TOTAL_CACHE_ACCESSES = tag lookups (128-byte lines) -- the L1 serves one per clock and compute unit --,
# TCP_TCC_READ_REQ = lookups that missed and went to the L2, TCP_GATE_EN2 = cycles the L1's core clock was enabled (busy), TA_TA_BUSY = cycles
# the addresser was busy, TA_TOTAL_WAVEFRONTS = vector-memory instructions it processed.
set -e
TAG=${1:-tcp}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_GATE_EN2_sum TCP_GATE_EN1_sum" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --roofline-steps 0 --steps 12 --warmup 2 > $OUT/bench_$i.log 2>&1 || true
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for fn in glob.glob('/tmp/pmc_${TAG}_*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if 'k_hash' not in k:
            continue
        k = k.replace('void ', '', 1).replace('(anonymous namespace)::', '').split('(')[0] + ' grid=' + r.get('Grid_Size', '?')
        a = agg[k][r['Counter_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
for fn in glob.glob('/tmp/pmc_${TAG}_1/*kernel_trace.csv'):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if 'k_hash' not in k:
            continue
        k = k.replace('void ', '', 1).replace('(anonymous namespace)::', '').split('(')[0] + ' grid=' + r.get('Grid_Size', r.get('Grid_Size_X', '?'))
        dur[k][0] += 1; dur[k][1] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3
names = sorted({n for d in agg.values() for n in d})
with open('$OUT/tcp_per_kernel.csv', 'w') as o:
    o.write('kernel,dispatches,avg_us_under_counters,' + ','.join('avg_' + n for n in names) + ',tag_lookups_per_CU_per_us,l2_requests_per_lookup\n')
    for k, d in sorted(agg.items()):
        avg = {n: (d[n][1] / d[n][0] if d[n][0] else float('nan')) for n in names}
        us = dur[k][1] / dur[k][0] if dur[k][0] else float('nan')
        acc, l2 = avg.get('TCP_TOTAL_CACHE_ACCESSES_sum', float('nan')), avg.get('TCP_TCC_READ_REQ_sum', float('nan'))
        o.write(f'"{k}",{max(v[0] for v in d.values())},{us:.1f},' + ','.join(f'{avg[n]:.1f}' for n in names) + f',{acc / 256 / us:.1f},{l2 / acc:.4f}\n')
print(open('$OUT/tcp_per_kernel.csv').read())
PY
