mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_fullsize.py -q -x -p no:cacheprovider 2>&1 | tail -8
for h in 0 1; do MFX_GRAM_HIP=$h timeout 300 python bench.py --mode train --no-cpu-baseline --steps 30 --repeats 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('train gram_hip=$h',d['value'],d['ms_per_step'],d['config']['loss_last_step'])"; done
MFX_GRAM_HIP=1 timeout 300 python bench.py --mode train --dtype fp16 --no-cpu-baseline --steps 30 --repeats 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('train fp16 gram_hip=1',d['value'],d['ms_per_step'],d['config']['loss_last_step'], d['config'].get('loss_scale'))"
