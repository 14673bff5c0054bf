#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c38; mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests/test_gpu_train_step.py -q -p no:cacheprovider -k "bitwise or sync_bn_collectives" 2>&1 | grep -E "^FAILED|passed|failed" >> $O/flaky.txt
done
timeout 900 python -m pytest tools/probes/syncbn_flaky_dbg.py -q -p no:cacheprovider -s -k "bitwise or zz_syncbn" 2>&1 | grep -E "state equal|eager step|replay loss|modules whose|passed|failed" > $O/dbg3.txt
timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/full.log 2>&1; tail -15 $O/full.log > $O/full_tail.txt
timeout 900 python bench.py --mode train --no-cpu-baseline 2>$O/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'], d['config'].get('loss_last_step'))" > $O/train.txt 2>&1
