export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for o in 1 0 1 0; do
MFX_CONV_STATS=$o timeout 600 python bench.py --mode train --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stats=$o', d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "conv_epilogue" 2>&1 | tail -1
