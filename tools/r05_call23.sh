#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_cw" 2>&1 | tail -4
  timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -4
  MFX_TRACE_CW=1 timeout 900 python bench.py --mode train --legs none 2> gpurun_out/cw_trace.err | tail -1 | cut -c1-330
  grep cw-fallback gpurun_out/cw_trace.err | sort | uniq -c | sort -rn ) > gpurun_out/cw_train.md 2>&1
cat gpurun_out/cw_train.md
