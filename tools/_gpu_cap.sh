mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_reference_golden.py -q -m gpu > gpurun_out/r03/t_golden.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r03/t_golden.log | tail -8
python tools/golden_margin.py > gpurun_out/r03/golden_margin.jsonl 2>gpurun_out/r03/golden_margin.err; grep -E "model_" gpurun_out/r03/golden_margin.jsonl | cut -c1-300
