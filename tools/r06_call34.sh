#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c34; mkdir -p $O
timeout 900 python bench.py --mode train --no-cpu-baseline 2>$O/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'], d['config'].get('loss_last_step'))" > $O/train.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x tests/test_gpu_train_step.py tests/test_gpu_kitti_encode.py > $O/t.log 2>&1; tail -15 $O/t.log > $O/t_tail.txt
