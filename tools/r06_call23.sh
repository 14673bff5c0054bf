#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c23; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -k "ext_dcn or dcn_split_vs_oracle or unsupported_geometry or test_gpu_dcn_surface or project_gemm" > $O/t.log 2>&1; tail -6 $O/t.log > $O/t_tail.txt
python tools/ext_cost.py 2>/dev/null | tail -1 > $O/ext_cost.txt
MFX_OPTIONS="ext_bwd_fast=0" python tools/ext_cost.py 2>/dev/null | tail -1 >> $O/ext_cost.txt
