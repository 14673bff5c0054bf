#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
( timeout 300 python tools/pointwise_bench.py 8 2>&1 | grep -v amdgpu | awk -F'|' '{printf "%s:%s; ", $2, $4}'; echo
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -2
  timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
  timeout 600 python bench.py --legs none --no-families --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" ) > gpurun_out/pointwise_final.md 2>&1
cat gpurun_out/pointwise_final.md
