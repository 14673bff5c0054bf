# round 5, GPU call 2: the default bench line with the new legs (b32, rotating train batches, family rooflines), streams 1 vs 2
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 900 python bench.py > gpurun_out/r05_c2_bench_default.json 2> gpurun_out/r05_c2_bench_default.err ) 2>&1 | grep real; echo rc=$?; tail -3 gpurun_out/r05_c2_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c2_bench_default.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])
for f in d['roofline_families']['families']: print('  ', f['family'], f['us_per_step'], f['frac'])
for k in ('b32','fp32_parity','fp16x2_parity','pipeline','fp16','train','train_fp16'):
    v=d.get(k,{})
    print(k, v.get('value'), v.get('ms_per_step'), v.get('error'), (v.get('config') or {}).get('timing'))
PY
for st in 1 2; do timeout 300 python bench.py --legs none --no-cpu-baseline --streams $st > gpurun_out/r05_c2_streams$st.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r05_c2_streams$st.json'));print('streams',$st,d['value'],d['ms_per_step'])"; done
