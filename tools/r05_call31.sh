#!/bin/bash
# tile / KC sweep of the 1x1 and Root launches (option conv_tile / cat_tile: 3 = 128x64, 4 = 64x64, 5 = 128x128, 6 = 64x128; kc = 4: 64-byte k-iterations)
cd /root/repo
mkdir -p gpurun_out
for o in "conv_tile=4,cat_tile=4" "conv_tile=4,cat_tile=4,kc=4" "conv_tile=3,cat_tile=3" "conv_tile=6,cat_tile=6" "kc=4" "conv_tile=5,cat_tile=5"; do
  echo "== $o"; timeout 300 python tools/pointwise_bench.py 8 $o 2>&1 | grep -v amdgpu | grep -v "^|---" | awk -F'|' '{printf "%s:%s->%s; ", $2, $4, $5}'; echo
done > gpurun_out/pointwise_tiles.md 2>&1
cat gpurun_out/pointwise_tiles.md
