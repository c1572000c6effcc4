#!/bin/bash
# round 6: the fused sweep (gather inside k_sdf_mlp2) -- parity, then same-box A/B against the two-launch form
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "sweep_with_the_gather or wide_sdf_sweep or fused_mfma_sdf" -s 2>&1 | grep -E "PARITY|passed|failed|Error|error" | tail -40 > $O/sweep_parity.txt
timeout 1200 python -m pytest tests/test_stock_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $O/sweep_parity.txt
bash tools/ab_env.sh HOLOSCENE_SDF_SWEEP_FUSED 0 1 3 > $O/sweep_ab.txt 2>&1
bash tools/exp/kstat_env.sh HOLOSCENE_SDF_SWEEP_FUSED 0 1 'k_hash_fwd_pair|k_sdf_mlp2|k_sampler' > $O/sweep_kstat.txt 2>&1
