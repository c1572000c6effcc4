F='PARITY convergence (rgb_loss|bf16|fp32)|passed|failed'
python -m pytest tests/test_hash_gpu.py -m gpu -q -x -k "reduce_and_step" > gpurun_out/r04_step_unit.log 2>&1; tail -15 gpurun_out/r04_step_unit.log
for i in 1 2 3; do HOLOSCENE_TABLE_STEP=0 python -m pytest tests/test_convergence_gpu.py -m gpu -q -s 2>&1 | grep -E "$F"; done > gpurun_out/r04_conv_step0.log 2>&1
for i in 1 2; do python -m pytest tests/test_convergence_gpu.py -m gpu -q -s 2>&1 | grep -E "$F|Error"; done > gpurun_out/r04_conv_step1.log 2>&1
cd .ab_old && for i in 1 2; do python -m pytest tests/test_convergence_gpu.py -m gpu -q -s 2>&1 | grep -E "$F"; done > ../gpurun_out/r04_conv_old.log 2>&1
cd ..; for f in step0 step1 old; do echo == $f; grep rgb_loss: gpurun_out/r04_conv_$f.log; done
