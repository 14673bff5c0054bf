#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c30; mkdir -p $O
timeout 1500 python tools/probes/infer_repeat.py > $O/infer_repeat.txt 2>&1
