#!/bin/bash
# profiles/run_pmc_issue.sh <tag> -- is a kernel bound by vector-instruction ISSUE?  ("show a counter at >= 0.8 of its rate")
# SQ instruction counters, each set in its own pass with --kernel-trace only, over the default bench's replayed graph:
#   SQ_INSTS_VALU (vector-ALU instructions issued, all waves), SQ_ACTIVE_INST_VALU (cycles a wave spent with a VALU instruction executing,
#   summed over waves), SQ_INST_CYCLES_VMEM / SQ_ACTIVE_INST_VMEM (the same for vector memory), SQ_WAIT_INST_ANY (wave-cycles spent
#   waiting on s_waitcnt), SQ_WAVE_CYCLES (wave-cycles resident), SQ_BUSY_CYCLES (summed over the 32 shader engines: / 32 = kernel cycles).
# A wave64 VALU instruction occupies its SIMD's 16-lane ALU for 4 cycles, so the issue roof is 1 instruction per 4 cycles and SIMD:
#   valu_issue_frac = 4 * SQ_INSTS_VALU / (1024 SIMDs * kernel cycles).
set -e
TAG=${1:-issue}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --roofline-steps 0 --steps 12 --warmup 2 > $OUT/bench_$i.log 2>&1 || true
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for fn in glob.glob('/tmp/pmc_${TAG}_*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if not any(t in k for t in ('k_hash', 'k_sampler', 'k_sdf_mlp2', 'k_rr_', 'k_appear2', 'k_wgrad', 'k_composite', 'k_trunk')):
            continue
        k = k.replace('void ', '', 1).replace('(anonymous namespace)::', '').split('(')[0] + ' grid=' + r.get('Grid_Size', '?')
        a = agg[k][r['Counter_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
names = sorted({n for d in agg.values() for n in d})
with open('$OUT/issue_per_kernel.csv', 'w') as o:
    o.write('kernel,dispatches,' + ','.join('avg_' + n for n in names) + ',kernel_cycles,valu_issue_frac,valu_active_over_wave_cycles,wait_over_wave_cycles,vmem_active_over_wave_cycles\n')
    for k, d in sorted(agg.items()):
        avg = {n: (d[n][1] / d[n][0] if n in d and d[n][0] else float('nan')) for n in names}
        g = lambda n: avg.get(n, float('nan'))
        cyc = g('SQ_BUSY_CYCLES') / 32.0
        wc = g('SQ_WAVE_CYCLES')
        o.write(f'"{k}",{max(v[0] for v in d.values())},' + ','.join(f'{avg[n]:.1f}' for n in names)
                + f',{cyc:.0f},{4 * g("SQ_INSTS_VALU") / (1024 * cyc):.4f},{g("SQ_ACTIVE_INST_VALU") / wc:.4f},{g("SQ_WAIT_INST_ANY") / wc:.4f},{g("SQ_ACTIVE_INST_VMEM") / wc:.4f}\n')
print(open('$OUT/issue_per_kernel.csv').read())
PY
