mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_kernels_vs_oracle.py tests/test_gpu_dcn_surface.py tests/test_gpu_e2e.py tests/test_gpu_split_precision.py -q -x -p no:cacheprovider 2>&1 | tail -3
python tools/dcn_layers_bench.py 8 3.0 2>&1 | grep -v amdgpu | cut -c1-80
for st in 1 2; do timeout 300 python bench.py --legs none --no-cpu-baseline --streams $st > gpurun_out/r05_c5_streams$st.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r05_c5_streams$st.json'));print('streams',$st,d['value'],d['ms_per_step']); [print('  ',f['family'],f['us_per_step'],f['frac']) for f in d['roofline_families']['families']]"; done
timeout 300 python bench.py --dtype fp16x2 --legs none --no-cpu-baseline --streams 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('fp16x2',d['value'],d['ms_per_step'])"
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 30 --repeats 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('train',d['value'],d['ms_per_step'])"
