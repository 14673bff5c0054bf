mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "stem or transposed_read" > gpurun_out/t1.log 2>&1; grep -E "passed|failed|FAILED|Error|assert" gpurun_out/t1.log | tail -8
timeout 600 python bench.py --mode train --no-cpu-baseline --steps 10 2>&1 | tail -1 | cut -c100-330
