python -m pytest tests/test_gpu_train.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
python bench.py --mode train --no-cpu-baseline --steps 5 2>&1 | tail -1 | cut -c90-200
