mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider -k "gram" 2>&1 | tail -15
