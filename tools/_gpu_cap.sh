mkdir -p gpurun_out/r03
python tools/ab_pow2_stride.py > gpurun_out/r03/ab_pow2_stride.jsonl 2>&1; cat gpurun_out/r03/ab_pow2_stride.jsonl | cut -c1-300
