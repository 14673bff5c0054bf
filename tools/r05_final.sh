# final check of the round: full GPU suite, smoke(), the driver's default bench command
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r05_full_gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r05_full_gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc=$?"; cut -c1-160 gpurun_out/r05_bench_default.json
timeout 600 python bench.py --dtype fp16x2 --no-cpu-baseline --legs none > gpurun_out/r05_bench_infer_fp16x2.json 2>/dev/null; cut -c1-160 gpurun_out/r05_bench_infer_fp16x2.json
