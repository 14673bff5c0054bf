python -m pytest tests/test_gpu_train.py -q -m gpu -p no:cacheprovider -k "focal or loss or train_steps or training_forward" 2>&1 | tail -12 | cut -c1-220
python bench.py --mode train --no-cpu-baseline --steps 5 2>&1 | tail -1 | cut -c90-200
