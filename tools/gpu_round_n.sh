mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py tests/test_gpu_ops.py -q > gpurun_out/t4.log 2>&1; grep -E "passed|failed|FAILED|Error" gpurun_out/t4.log | head -12
