#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c42; mkdir -p $O; rm -f $O/train_ab2.txt
for v in "wgrad_min_m=128" "wgrad_min_m=64" "wgrad_min_m=32" "wgrad_min_m=128,wgrad_blocks=1200" "wgrad_min_m=128,wgrad_blocks=300" "wgrad_min_m=128,wgrad_ws_blocks=2400" "wgrad_min_m=128"; do
  echo "## $v" >> $O/train_ab2.txt
  MFX_OPTIONS=$v timeout 600 python bench.py --mode train --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'], d['config'].get('loss_last_step'))" >> $O/train_ab2.txt
done
