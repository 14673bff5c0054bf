# final check of the round: full GPU suite, smoke()
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r05_full_gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r05_full_gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
