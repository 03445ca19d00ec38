mkdir -p gpurun_out/r03
(time timeout 900 python tools/shard_plan_build.py) > gpurun_out/r03/shard_plan_build.jsonl 2> gpurun_out/r03/shard_plan_build.err
cat gpurun_out/r03/shard_plan_build.jsonl; tail -5 gpurun_out/r03/shard_plan_build.err
