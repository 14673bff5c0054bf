#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c29; mkdir -p $O
for v in e3; do
  export MFX_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/lib_$v.so
  RAW0=1 timeout 900 python tools/probes/dcn_bwd_repeat.py 2>&1 | grep "^form" > $O/rep_$v.txt
done
