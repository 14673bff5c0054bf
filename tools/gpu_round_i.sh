for i in 1 2; do python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py -q -m gpu -p no:cacheprovider 2>&1 | grep "^FAILED\|passed\|failed\|^E  " | head -8 | cut -c1-250; done
