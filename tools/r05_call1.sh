# round 5, GPU call 1: parity of the new 3x3 kernel + activation rewrite, A/B table, inference step with and without it, step timeline
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_kernels_vs_oracle.py -q -x -p no:cacheprovider > gpurun_out/r05_c1_tests.log 2>&1; tail -4 gpurun_out/r05_c1_tests.log
timeout 300 python tools/conv_cw_bench.py 8 > gpurun_out/r05_conv_cw_ab.md 2>&1; cat gpurun_out/r05_conv_cw_ab.md
for cw in 0 1; do timeout 300 python bench.py --legs none --no-cpu-baseline --opts halo_cw=$cw > gpurun_out/r05_c1_bench_cw$cw.json 2>gpurun_out/r05_c1_bench_cw$cw.err; cut -c1-160 gpurun_out/r05_c1_bench_cw$cw.json; echo; done
bash tools/profile_infer_step.sh > /dev/null 2>&1; for f in gpurun_out/r04_b_inference_*; do mv $f ${f/r04_b/r05_c1}; done; head -40 gpurun_out/r05_c1_inference_replay_kernel_timeline.md | cut -c1-120
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -p no:cacheprovider > gpurun_out/r05_c1_e2e.log 2>&1; tail -3 gpurun_out/r05_c1_e2e.log
