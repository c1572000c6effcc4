#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu -k "eight_ranks" -s 2>&1 | grep -E "PARITY|passed|failed|Error|error|assert" | tail -40 > $O/ws8.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "configs3" -s 2>&1 | grep -E "PARITY|passed|failed|Error|error|assert" | tail -20 > $O/c3.txt
timeout 900 python bench.py --steps 60 --no-fp32-point --no-trajectory-point --cpu-seconds 4 --roofline-counters > $O/bench_counters.json 2> $O/bench_counters.err
timeout 600 python bench.py --steps 60 --objects 21 --img-res 584 876 --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point > $O/bench_c3.json 2> $O/bench_c3.err
