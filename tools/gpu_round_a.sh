mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_bf16_kernels_vs_oracle.py tests/test_gpu_e2e.py tests/test_gpu_train_step.py -q -m gpu -s -p no:cacheprovider > gpurun_out/t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/t1.log
timeout 600 python bench.py > gpurun_out/r02_a_bench_infer.json 2> gpurun_out/r02_a_bench_infer.err
timeout 300 python bench.py --dtype fp32 --no-cpu-baseline > gpurun_out/r02_a_bench_infer_fp32.json 2> gpurun_out/r02_a_bench_infer_fp32.err
timeout 300 python bench.py --batch 32 --no-cpu-baseline --steps 10 > gpurun_out/r02_a_bench_infer_b32.json 2> gpurun_out/r02_a_bench_infer_b32.err
timeout 900 python bench.py --mode train > gpurun_out/r02_a_bench_train.json 2> gpurun_out/r02_a_bench_train.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o r02a -- python $GRAFT_REPO_ROOT/bench.py --mode train --no-cpu-baseline --steps 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof_train -name "*.db" | head -1); python tools/prof_summary.py $DB 5 > gpurun_out/r02_a_train_step_kernel_stats.md 2>> gpurun_out/prof_train.log
tail -3 gpurun_out/t1.log; cat gpurun_out/r02_a_bench_infer.json | cut -c1-600; cat gpurun_out/r02_a_bench_train.json | cut -c1-600; tail -2 gpurun_out/r02_a_bench_train.err
