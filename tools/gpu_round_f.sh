python -m pytest tests/test_gpu_train.py -q -m gpu -p no:cacheprovider -k "wgrad_transposed or dcn_tile or conv_grads_bf16" 2>&1 | tail -12 | cut -c1-300
for b in 256 512 1024 2048; do python bench.py --mode train --no-cpu-baseline --steps 5 --opts wgrad_tr_blocks=$b 2>&1 | tail -1 | cut -c90-200; done
python bench.py --mode train --no-cpu-baseline --steps 5 --opts wgrad_tr=0 2>&1 | tail -1 | cut -c90-200
