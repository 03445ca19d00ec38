mkdir -p gpurun_out/r03
export TFGX_BENCH_BACKEND=gloo
(time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1) > gpurun_out/r03/bench_2rank_gloo.json 2> gpurun_out/r03/bench_2rank_gloo.err
tail -3 gpurun_out/r03/bench_2rank_gloo.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_2rank_gloo.json').read().strip().splitlines()[-1])
print(d["config"], d["value"], d["ms_per_step"], d["plan_build_s"])
for r in d["roofline"]["per_rank"]: print(r)
PY
unset TFGX_BENCH_BACKEND
(time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/r03/tests_all.log 2>&1
tail -8 gpurun_out/r03/tests_all.log
