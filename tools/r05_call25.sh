#!/bin/bash
# split-precision range sentinel: tests + cost (same box: HEAD vs the library before the change)
cd /root/repo
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "sentinel" 2>&1 | tail -3
  
  for v in head nocheck head nocheck; do
    if [ $v = nocheck ]; then export MFX_LIB_PATH=/root/repo/build_variants/lib_nocheck.so; else unset MFX_LIB_PATH; fi
    echo "== $v"; timeout 600 python bench.py --dtype fp16x2 --legs none --no-families --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done ) > gpurun_out/sentinel.md 2>&1
cat gpurun_out/sentinel.md
