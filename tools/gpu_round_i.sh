python -m pytest tests/test_gpu_train.py -q -m gpu -p no:cacheprovider -k "conv_grads or wgrad or dcn_tile" 2>&1 | tail -2 | cut -c1-220
python bench.py --mode train --no-cpu-baseline --steps 5 2>&1 | tail -1 | cut -c90-200
