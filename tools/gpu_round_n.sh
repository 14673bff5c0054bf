for o in 2 4 1 2; do
timeout 600 python bench.py --no-cpu-baseline --streams $o 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams=$o', d['value'], d['ms_per_step'], d['config'].get('detections_last_step'), d['config'].get('vs_reference',{}).get('max_abs_dlogit'))"
done
timeout 600 python bench.py --no-cpu-baseline --streams 2 --batch 32 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b32 streams=2', d['value'], d['ms_per_step'])"
