bash profiles/run_rocprof.sh it --steps 30 --warmup 15 >/dev/null 2>&1
T=$(find /tmp/prof_it -name "*kernel_trace.csv" | head -1)
python tools/iter_kernels.py $T 16
