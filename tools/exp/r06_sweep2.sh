#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "sweep_with_the_gather" -s 2>&1 | grep -E "PARITY|passed|failed|Error|error" | tail -40 > $O/sweep2_parity.txt
(HOLOSCENE_SWEEP_STAGGER=0 ./tools/exp/sweep_prof 1024; HOLOSCENE_SWEEP_STAGGER=4 ./tools/exp/sweep_prof 1024; HOLOSCENE_SWEEP_STAGGER=0 ./tools/exp/sweep_prof 4096) > $O/sweep2_prof.txt 2>&1
bash tools/exp/kstat_envs.sh HOLOSCENE_SWEEP_STAGGER "k_sdf_mlp2|k_sampler_update" 0 4 8 > $O/sweep2_kstat.txt 2>&1
