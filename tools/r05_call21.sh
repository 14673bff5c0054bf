#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python bench.py --legs none --no-families 2>/dev/null | tail -1 > gpurun_out/bench_infer_quick.json
python - <<'PY'
import json
d = json.load(open('/root/repo/gpurun_out/bench_infer_quick.json'))
print(d['value'], d['ms_per_step'], d.get('roofline'))
PY
