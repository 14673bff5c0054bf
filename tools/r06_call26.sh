#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c26; mkdir -p $O
timeout 900 python tools/probes/train_torch_dispatch.py > $O/torch_dispatch.txt 2>&1
