set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_k -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --steps 60 --roofline-steps 2 > $OUT/bench.log 2>&1 || true
TR=$(find /tmp/prof_k -name "*kernel_trace.csv" | head -1)
for k in 40 41 42 43 44 45 46 47 48 49 50; do python $GRAFT_REPO_ROOT/tools/iter_kernels.py $TR $k > $OUT/it$k.txt 2>&1 || true; head -2 $OUT/it$k.txt | tail -1; done
