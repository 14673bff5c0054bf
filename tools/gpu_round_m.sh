mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py -x -q -k "graphed or conv or checkpoint" > gpurun_out/t2.log 2>&1; grep -E "passed|failed|FAILED|Error|assert" gpurun_out/t2.log | tail -6
timeout 600 python bench.py --mode train --no-cpu-baseline --steps 10 2>&1 | tail -1 | cut -c100-330
