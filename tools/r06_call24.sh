#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c24; mkdir -p $O
for v in 1 0 1 0; do
  echo "## MFX_WGRAD_STREAM=$v" >> $O/train_ab.txt
  MFX_WGRAD_STREAM=$v timeout 600 python bench.py --mode train --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'], d['config'].get('loss_last_step'))" >> $O/train_ab.txt
done
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x tests/test_gpu_train_step.py tests/test_gpu_train.py > $O/t.log 2>&1; tail -5 $O/t.log > $O/t_tail.txt
