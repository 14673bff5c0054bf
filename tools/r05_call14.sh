#!/bin/bash
# corner-quad DCN kernel (LDS-DMA form): correctness vs the other kernels + time per layer
cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/dcn_cq_probe.py 8 3.0 bf16 3 2>&1 | grep -v amdgpu.ids > gpurun_out/dcn_cq_probe.md
cat gpurun_out/dcn_cq_probe.md
