#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c21; mkdir -p $O
timeout 300 python tools/dcn_ps_bench.py 8 "" > $O/ps_bench.md 2> $O/ps_bench.err
timeout 300 python tools/dcn_ps_check.py > $O/check.txt 2>&1
