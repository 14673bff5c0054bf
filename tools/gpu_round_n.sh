timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "packed_conv_operands" 2>&1 | tail -3
