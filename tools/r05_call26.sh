#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "sentinel" 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --dtype fp16x2 --legs none --no-families --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('operand_range_ok'))"; done
