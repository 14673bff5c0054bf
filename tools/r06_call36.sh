#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c36; mkdir -p $O
GRAPH=1 timeout 1500 python tools/probes/train_forward_repeat.py > $O/fwd_replay.txt 2>&1
