#!/bin/bash
# profiles/run_final_r06.sh -- the round's closing measurements in one gpurun call (HS_COMMIT=<hash> names the commit):
# (1) rocprofv3 --kernel-trace --stats of the default bench's hot loop: per-kernel summary + one regular and one background-patch iteration in
# launch order; (2) the two PMC traffic passes + calibration -> pmc_traffic.json; (3) SQ passes (matrix pipe busy, LDS conflicts); (4) vector-L1 /
# addresser passes of the hash kernels; (5) SQ issue passes (VALU / VMEM / wait) -> issue_per_kernel.csv and, with (4), gather_bound.json: every
# candidate roof of the gather sweep as a fraction of its own rate; (6) the GPU test suite's PARITY lines; (7) the full default bench line.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 200 --warmup 20 > $OUT/bench_under_rocprof.log 2>&1 || true
find /tmp/prof_f -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
TR=$(find /tmp/prof_f -name "*kernel_trace.csv" | head -1)
for k in 100 101 102 103 104 105 106 107 108 109; do python $GRAFT_REPO_ROOT/tools/iter_kernels.py $TR $k > /tmp/iter_$k.txt 2>&1 || true; done
# ten consecutive iterations: nine regular ones and the background-patch one (the longest span)
BG=$(for k in 100 101 102 103 104 105 106 107 108 109; do echo "$(grep -o "span [0-9.]*" /tmp/iter_$k.txt | cut -d" " -f2) $k"; done | sort -n | tail -1 | cut -d" " -f2)
RG=$(for k in 100 101 102 103 104 105 106 107 108 109; do echo "$(grep -o "span [0-9.]*" /tmp/iter_$k.txt | cut -d" " -f2) $k"; done | sort -n | sed -n 5p | cut -d" " -f2)
cp /tmp/iter_$BG.txt $OUT/bg_iteration_kernels.txt; cp /tmp/iter_$RG.txt $OUT/iteration_kernels.txt
tail -1 $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json || true
cd $GRAFT_REPO_ROOT
timeout 900 bash profiles/run_pmc_r02.sh r06 > $OUT/pmc.log 2>&1 || true
cp gpurun_out/pmc_r06/pmc_traffic.json gpurun_out/pmc_r06/pmc_per_kernel.csv $OUT/ 2>/dev/null || true
timeout 900 bash profiles/run_pmc_sq.sh r06sq > $OUT/pmc_sq.log 2>&1 || true
cp gpurun_out/pmc_r06sq/sq_per_kernel.csv $OUT/pmc_sq_per_kernel.csv 2>/dev/null || true
timeout 900 bash profiles/run_pmc_tcp.sh r06tcp > $OUT/pmc_tcp.log 2>&1 || true
cp gpurun_out/pmc_r06tcp/tcp_per_kernel.csv $OUT/pmc_tcp_hash_kernels.csv 2>/dev/null || true
timeout 1200 bash profiles/run_pmc_issue.sh r06issue > $OUT/pmc_issue.log 2>&1 || true
cp gpurun_out/pmc_r06issue/issue_per_kernel.csv $OUT/pmc_issue_per_kernel.csv 2>/dev/null || true
python tools/pmc_gather_bound.py $OUT/pmc_tcp_hash_kernels.csv $OUT/pmc_issue_per_kernel.csv $OUT/gather_bound.json > /dev/null 2>&1 || true
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -a "PARITY\|passed\|failed" > $OUT/parity_measured.txt || true
tail -1 $OUT/parity_measured.txt
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err || true
tail -c 700 $OUT/bench.json
# (8) round 6: per-kernel cost at three ray counts (fixed vs per-ray), the other configurations, the in-run addresser counter, K = 64
rm -f $OUT/kstat_rays.txt
for rays in 512 1024 4096; do echo "== --rays $rays" >> $OUT/kstat_rays.txt; bash tools/exp/kstat_args.sh --rays $rays >> $OUT/kstat_rays.txt 2>&1; done
python tools/fixed_vs_per_ray.py $OUT/kstat_rays.txt > $OUT/fixed_vs_per_ray.txt 2>&1 || true
B="python bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point"
for cfg in "--rays 512" "--rays 1024 --objects 21 --img-res 584 876" "--rays 4096 --steps 100" "--rays 2048 --samples 192 --steps 100" "--rays 1024 --objects 64" "--rays 1024 --objects 40" "--rays 256 --samples 64 --objects 2 --levels 8 --end-size 256 --logmap 15"; do
  echo "== $cfg" >> $OUT/other_config_shapes.txt
  $B $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['iteration_kinds'])" >> $OUT/other_config_shapes.txt 2>&1
done
timeout 900 python bench.py --steps 60 --no-fp32-point --no-trajectory-point --cpu-seconds 4 --roofline-counters > $OUT/bench_roofline_counters.json 2> $OUT/bench_roofline_counters.err || true
bash tools/exp/kstat_args.sh --objects 64 > $OUT/kstat_k64.txt 2>&1 || true
git rev-parse HEAD > $OUT/commit.txt 2>/dev/null || echo "${HS_COMMIT:-unknown}" > $OUT/commit.txt
