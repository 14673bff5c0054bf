mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py -x -q -k "dcn or network or train_steps or graphed or oracle" > gpurun_out/t7.log 2>&1; grep -E "passed|failed|FAILED|Error|assert " gpurun_out/t7.log | head -8
for i in 1 2; do timeout 300 python bench.py --mode train --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
