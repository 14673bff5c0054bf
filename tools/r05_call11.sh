mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_kernels_vs_oracle.py tests/test_gpu_e2e.py -q -x -p no:cacheprovider 2>&1 | tail -4
for cw in 0 1; do timeout 300 python bench.py --legs none --no-cpu-baseline --opts halo_cw=$cw > gpurun_out/r05_c11_$cw.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r05_c11_$cw.json'));print('bf16 cw=$cw',d['value'],d['ms_per_step']); [print('  ',f['family'],f['us_per_step'],f['frac']) for f in d['roofline_families']['families']]"; done
