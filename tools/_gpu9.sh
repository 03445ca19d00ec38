mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_backward.py -q -x -k "sddmm or grad_x_and_w or registered" 2>&1 | tail -3
python tools/bench_sweep.py --only=backward > gpurun_out/r03/backward_a.jsonl 2> gpurun_out/r03/backward_a.err
cat gpurun_out/r03/backward_a.jsonl | cut -c1-600
