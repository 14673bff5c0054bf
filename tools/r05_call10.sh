mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_kernels_vs_oracle.py -q -x -p no:cacheprovider -k "cw or conv3x3 or halo" 2>&1 | tail -4
python tools/dcn_layers_bench.py 8 3.0 2>&1 | grep -v amdgpu | cut -c1-80
timeout 300 python bench.py --legs none --no-cpu-baseline > gpurun_out/r05_c10.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r05_c10.json'));print('bf16',d['value'],d['ms_per_step']); [print('  ',f['family'],f['us_per_step'],f['frac']) for f in d['roofline_families']['families']]"
