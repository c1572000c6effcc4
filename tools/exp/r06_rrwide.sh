#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_trunk_rr_gpu.py -x -q -m gpu -s 2>&1 | grep -E "PARITY rr wide|passed|failed|Error|error|assert" | tail -40 > $O/rrwide_unit.txt
timeout 1500 python -m pytest tests/test_stock_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "k40 or k64 or wide or K64 or 64" 2>&1 | tail -15 > $O/rrwide_fixtures.txt
bash tools/ab_env.sh HOLOSCENE_RR_WIDE 0 1 2 --objects 64 > $O/rrwide_ab64.txt 2>&1
bash tools/ab_env.sh HOLOSCENE_RR_WIDE 0 1 1 --objects 40 > $O/rrwide_ab40.txt 2>&1
