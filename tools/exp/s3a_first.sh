mkdir -p gpurun_out/s3a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/s3a/gputest.txt
B="python bench.py --no-cpu-baseline --no-second-point --no-fp32-point --no-trajectory-point --steps 200 --warmup 20 --roofline-steps 0"
for i in 1 2 3; do
 for z in 1 0; do HOLOSCENE_PROLOGUE_ZERO=$z $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('zero=$z', d['ms_per_step'], d.get('ms_per_step_median'), round(d['value']))" >> gpurun_out/s3a/ab_zero.txt; done
done
cat gpurun_out/s3a/gputest.txt gpurun_out/s3a/ab_zero.txt
