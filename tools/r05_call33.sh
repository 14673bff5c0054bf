#!/bin/bash
# forked-stream proj DCN modules inside the captured step (MFX_PARALLEL_PROJ=1), re-measured with the round-5 kernels
cd /root/repo
for v in 0 1 0 1; do
  echo -n "MFX_PARALLEL_PROJ=$v: "; MFX_PARALLEL_PROJ=$v timeout 600 python bench.py --legs none --no-families --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
