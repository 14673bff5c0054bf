#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c39; mkdir -p $O
timeout 300 python tools/probes/adamw_capture_dbg.py 2>&1 | grep -v "^  File\|amdgpu.ids" | head -30 > $O/dbg.txt
