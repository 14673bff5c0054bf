#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c32; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_optim.py -q -p no:cacheprovider -x > $O/t_optim.log 2>&1; tail -25 $O/t_optim.log > $O/t_optim_tail.txt
for v in 1 0 1 0; do
  echo "## MFX_HIP_ADAMW=$v" >> $O/train_ab.txt
  MFX_HIP_ADAMW=$v timeout 600 python bench.py --mode train --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'], d['config'].get('loss_last_step'))" >> $O/train_ab.txt
done
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x tests/test_gpu_train_step.py > $O/t.log 2>&1; tail -15 $O/t.log > $O/t_tail.txt
