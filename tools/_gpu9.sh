mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_backward.py -q -x -k "sddmm or grad_x_and_w or registered or hub" 2>&1 | tail -3
python tools/ab_backward_ratios.py 2>/dev/null | tee gpurun_out/r03/backward_ratios.json | cut -c1-600
