python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c60-200
