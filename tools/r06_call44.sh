#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RTAG=r06 bash tools/profile_infer_step.sh fp16x2 f1_fused fp16x2_inference > gpurun_out/r06_fp16x2_prof.log 2>&1
