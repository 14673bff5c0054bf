#!/bin/bash
# fp16x2: where did 5 % go?  same box: HEAD / sentinel compiled out / sentinel out + scalar blend in the DCN gather loaders
cd /root/repo
for v in head nocheck scalar head nocheck scalar; do
  if [ $v = head ]; then unset MFX_LIB_PATH; else export MFX_LIB_PATH=/root/repo/build_variants/lib_$v.so; fi
  echo -n "$v: "; timeout 600 python bench.py --dtype fp16x2 --legs none --no-families --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
