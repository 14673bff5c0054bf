mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "loss" 2>&1 | tail -15
timeout 600 python bench.py --mode train --no-cpu-baseline --steps 10 2>&1 | tail -1 | cut -c1-330
