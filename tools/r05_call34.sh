#!/bin/bash
cd /root/repo
for o in "" "topk_strips=16" "topk_strips=32" "topk_strips=4" "topk_strips=12" ""; do
  echo -n "${o:-(defaults)}: "; MFX_OPTIONS="$o" timeout 300 python bench.py --legs none --no-cpu-baseline --no-families 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
