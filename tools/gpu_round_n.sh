timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c100-260
timeout 600 python bench.py --mode train --no-cpu-baseline --steps 10 2>&1 | tail -1 | cut -c100-260
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -2
