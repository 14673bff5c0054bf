#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c5; mkdir -p $O
for opt in "" "conv_tile=3" "conv_tile=5" "conv_tile=6" "conv_tile=7"; do
  echo "## opts: $opt" >> $O/ps_bench.md
  timeout 300 python tools/dcn_ps_bench.py 8 "$opt" >> $O/ps_bench.md 2>> $O/ps_bench.err
done
