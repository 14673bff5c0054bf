#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c48; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_gpu_train.py -q -p no:cacheprovider -k "bn_" 2>&1 | tail -25 > $O/t_bn.txt
for v in "bn_onepass=0" "bn_onepass=1" "bn_onepass=1,bn_onepass_grid=256" "bn_onepass=0" "bn_onepass=1"; do
  echo "## $v" >> $O/train_ab.txt
  MFX_OPTIONS=$v timeout 600 python bench.py --mode train --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'], d['config'].get('loss_last_step'))" >> $O/train_ab.txt
done
timeout 300 python tools/bn_bench.py > $O/bn_bench_onepass.txt 2>&1; timeout 300 python tools/bn_bench.py --opts bn_onepass=0 > $O/bn_bench_twolaunch.txt 2>&1
