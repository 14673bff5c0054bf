#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c43; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py -q -p no:cacheprovider 2>&1 | tail -6 > $O/t.txt
timeout 600 python bench.py --mode train --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'], d['config'].get('loss_last_step'))" > $O/train.txt
