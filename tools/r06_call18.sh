#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_c18
RTAG=r06 bash tools/profile_infer_step.sh fp16x2 f1_fused fp16x2_inference > gpurun_out/r06_c18/log1.txt 2>&1
RTAG=r06 bash tools/profile_infer_step.sh bf16 f1_fused b_inference > gpurun_out/r06_c18/log2.txt 2>&1
ls gpurun_out/ | grep r06_ > gpurun_out/r06_c18/ls.txt
