#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c40; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py -q -p no:cacheprovider -k "survives_other_models" 2>&1 | tail -5 > $O/t.txt
timeout 900 python tools/probes/train_torch_dispatch.py > $O/torch_dispatch.txt 2>&1
timeout 1200 python bench.py --legs none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('vs_reference'))" > $O/infer.txt 2>&1
