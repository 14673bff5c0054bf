#!/bin/bash
# packed-fp32 blend + 32-bit offsets in the gather loaders: per-layer A/B against the previous library on the same box
cd /root/repo
mkdir -p gpurun_out
( echo "== HEAD"; timeout 300 python tools/dcn_layers_bench.py 8 3.0 2>&1 | grep -v amdgpu.ids
  echo "== previous library"; MFX_LIB_PATH=/root/repo/build_variants/lib_prev.so timeout 300 python tools/dcn_layers_bench.py 8 3.0 2>&1 | grep -v amdgpu.ids
  echo "== tests"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_kernels_vs_oracle.py -x -q -k "dcn" 2>&1 | tail -5 ) > gpurun_out/dcn_pk_ab.md 2>&1
cat gpurun_out/dcn_pk_ab.md
