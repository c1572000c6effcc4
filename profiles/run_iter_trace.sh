set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-point --no-fp32-point --steps 60 --roofline-steps 2 > $OUT/bench.log 2>&1 || true
find /tmp/prof_k -name "*stats*.csv" -exec cp {} $OUT/ \;
TR=$(find /tmp/prof_k -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/iter_kernels.py $TR 40 > $OUT/iter40.txt 2>&1 || true
tail -2 $OUT/bench.log | cut -c1-200
