#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c51; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/full.log 2>&1; tail -15 $O/full.log > $O/full_tail.txt
