#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c45; mkdir -p $O
for v in 24 40 24 40 24 40; do
  echo "## MFX_DCN_PS_MAX_MB=$v" >> $O/ab.txt
  MFX_DCN_PS_MAX_MB=$v timeout 600 python bench.py --legs none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt
done
