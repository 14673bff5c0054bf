mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_dcn_surface.py tests/test_gpu_ops.py -q -x -p no:cacheprovider 2>&1 | tail -6
python tools/ext_cost.py 2>&1 | tail -6
