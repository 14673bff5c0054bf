python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
for b in 128 256 512; do python bench.py --mode train --no-cpu-baseline --steps 5 --opts wgrad_tr_blocks=$b 2>&1 | tail -1 | cut -c90-200; done
for b in 300 600; do python bench.py --mode train --no-cpu-baseline --steps 5 --opts wgrad_ws_blocks=$b 2>&1 | tail -1 | cut -c90-200; done
