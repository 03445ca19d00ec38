mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_backward.py tests/test_gpu_regressions.py tests/test_gpu_layers.py -q -m gpu -x > gpurun_out/r03/t_cap.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r03/t_cap.log | tail -8
python examples/demo_gcn.py --steps 200 2>&1 | tail -3
python examples/demo_gcn.py --steps 200 --hipgraph 2>&1 | tail -3
python bench.py --workload arxiv --extras --no-cpu-baseline --no-rmat > gpurun_out/r03/bench_arxiv_cap.json 2> gpurun_out/r03/bench_arxiv_cap.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_arxiv_cap.json').read().strip().splitlines()[-1])['extras']
print({k:v for k,v in d.items() if 'train' in k})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_ts -o ts -- python $GRAFT_REPO_ROOT/tools/profile_train_step.py gcn arxiv > /tmp/ts.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py /tmp/prof_ts/ts_results.db > gpurun_out/r03/train_step_arxiv_rocprof.md 2>&1
head -12 gpurun_out/r03/train_step_arxiv_rocprof.md | cut -c1-200
