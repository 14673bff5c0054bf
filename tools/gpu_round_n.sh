timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "bn" 2>&1 | grep -E "Error|assert|FAILED|passed|failed" | head -12
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py -x -q -k "network or graphed or train_steps or oracle or ddp" > gpurun_out/t3.log 2>&1; grep -E "passed|failed|FAILED" gpurun_out/t3.log | tail -3
