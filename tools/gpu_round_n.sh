timeout 600 python bench.py --mode train --no-cpu-baseline --steps 10 2>&1 | tail -1 | cut -c100-260
