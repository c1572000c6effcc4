#!/bin/bash
# profiles/run_pmc_sq.sh <tag> -- SQ counter passes (separate runs, --kernel-trace only) for the hand-written matrix-core kernels:
# how busy the MFMA pipe is while the kernel runs, and how much of the LDS time goes to bank conflicts.
set -e
TAG=${1:-sq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --roofline-steps 0 --steps 12 --warmup 2 > $OUT/bench_$i.log 2>&1 || true
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for fn in glob.glob('/tmp/pmc_${TAG}_*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if not any(t in k for t in ('k_sdf_mlp', 'k_sdf_mlp2', 'k_assemble', 'k_iter_', 'k_pack_iteration', 'k_draw_gather', 'k_sampler_draw', 'k_trunk_fwd', 'k_trunk_bwd', 'k_appear', 'k_rr_', 'k_wgrad', 'k_hash_fwd', 'k_hash_bwd_jac', 'k_hash_bin_reduce', 'k_sampler_update', 'k_composite')):
            continue
        k = k.replace('void ', '', 1).replace('(anonymous namespace)::', '').split('(')[0]
        a = agg[k][r['Counter_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
names = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_LDS', 'SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_WAVE_CYCLES']
with open('$OUT/sq_per_kernel.csv', 'w') as o:
    o.write('kernel,dispatches,' + ','.join('avg_' + n for n in names) + ',mfma_busy_over_sq_busy,lds_conflict_over_lds_active\n')
    for k, d in sorted(agg.items()):
        avg = {n: (d[n][1] / d[n][0] if d[n][0] else float('nan')) for n in names}
        n_disp = max(v[0] for v in d.values())
        r1 = avg['SQ_VALU_MFMA_BUSY_CYCLES'] / avg['SQ_BUSY_CYCLES'] if avg['SQ_BUSY_CYCLES'] else float('nan')
        r2 = avg['SQ_LDS_BANK_CONFLICT'] / avg['SQ_ACTIVE_INST_LDS'] if avg['SQ_ACTIVE_INST_LDS'] else float('nan')
        o.write(f'"{k}",{n_disp},' + ','.join(f'{avg[n]:.1f}' for n in names) + f',{r1:.4f},{r2:.4f}\n')
print(open('$OUT/sq_per_kernel.csv').read())
PY
