export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py -x -q -k "conv_epilogue or halo or conv2d" 2>&1 | tail -1
for o in 1 0 1; do
MFX_CONV_STATS=$o timeout 600 python bench.py --mode train --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stats=$o', d['ms_per_step'])"
done
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('infer', d['value'], d['ms_per_step'])"
