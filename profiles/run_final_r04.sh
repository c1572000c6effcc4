#!/bin/bash
# profiles/run_final_r04.sh -- the round's closing measurements in one gpurun call (HS_COMMIT=<hash> in the environment names the commit):
# (1) rocprofv3 --kernel-trace --stats of the default bench command's hot loop: per-kernel summary + one iteration's kernel sequence;
# (2) the two PMC traffic passes (FETCH_SIZE, WRITE_SIZE) + calibration -> pmc_traffic.json; (3) the SQ passes (matrix pipe busy, LDS bank
# conflicts); (4) the vector-L1 / addresser passes of the hash kernels; (5) the full default bench line (cpu_baseline, second point, fp32 point,
# trajectory point).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 200 --warmup 20 > $OUT/bench_under_rocprof.log 2>&1 || true
find /tmp/prof_f -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
TR=$(find /tmp/prof_f -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/iter_kernels.py $TR 100 > $OUT/iteration_kernels.txt 2>&1 || true
tail -1 $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json || true
cd $GRAFT_REPO_ROOT
timeout 900 bash profiles/run_pmc_r02.sh r04 > $OUT/pmc.log 2>&1 || true
cp gpurun_out/pmc_r04/pmc_traffic.json gpurun_out/pmc_r04/pmc_per_kernel.csv $OUT/ 2>/dev/null || true
timeout 900 bash profiles/run_pmc_sq.sh r04sq > $OUT/pmc_sq.log 2>&1 || true
cp gpurun_out/pmc_r04sq/sq_per_kernel.csv $OUT/pmc_sq_per_kernel.csv 2>/dev/null || true
timeout 900 bash profiles/run_pmc_tcp.sh r04tcp > $OUT/pmc_tcp.log 2>&1 || true
cp gpurun_out/pmc_r04tcp/tcp_per_kernel.csv $OUT/pmc_tcp_hash_kernels.csv 2>/dev/null || true
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err || true
tail -c 600 $OUT/bench.json
