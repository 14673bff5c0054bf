#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
for v in cq_nln; do
  echo "== variant: ${v:-head} (first form, dcn_cq=2)"
  export MFX_LIB_PATH=/root/repo/build_variants/lib_$v.so
  timeout 300 python tools/dcn_cq_probe.py 8 3.0 bf16 2 2>&1 | grep -v amdgpu.ids
done > gpurun_out/dcn_cq_variants2.md 2>&1
cat gpurun_out/dcn_cq_variants2.md
