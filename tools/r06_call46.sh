#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c46; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_train.py -q -p no:cacheprovider -k "dcn" 2>&1 | tail -4 > $O/t_dcn.txt
for v in "dcn_bt_fly_bias=1" "dcn_bt_fly_bias=0" "dcn_bt_fly_bias=1" "dcn_bt_fly_bias=0"; do
  echo "## $v" >> $O/train_ab.txt
  MFX_OPTIONS=$v timeout 600 python bench.py --mode train --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'], d['config'].get('loss_last_step'))" >> $O/train_ab.txt
done
