# last refresh of the round: default bench line + inference step timeline at the final HEAD (after the pointwise-layer tweaks)
export RTAG=r05; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; cut -c1-200 gpurun_out/r05_bench_default.json; echo
timeout 600 python bench.py --batch 32 --no-cpu-baseline --legs none > gpurun_out/r05_bench_infer_b32.json 2>/dev/null; cut -c1-160 gpurun_out/r05_bench_infer_b32.json; echo
timeout 600 python bench.py --dtype fp16x2 --no-cpu-baseline --legs none > gpurun_out/r05_bench_infer_fp16x2.json 2>/dev/null; cut -c1-160 gpurun_out/r05_bench_infer_fp16x2.json; echo
timeout 900 python bench.py --mode train > gpurun_out/r05_bench_train.json 2> gpurun_out/r05_bench_train.err; cut -c1-200 gpurun_out/r05_bench_train.json; echo
cd /tmp && rm -rf /tmp/prof_inf && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o prof -- python $R/bench.py --legs none --no-cpu-baseline --no-families --steps 20 --repeats 1 > $R/gpurun_out/prof_inf.log 2>&1
cd $R; python tools/prof_summary.py $(find /tmp/prof_inf -name "*.db" | head -1) > gpurun_out/r05_b_inference_kernel_stats.md; head -6 gpurun_out/r05_b_inference_kernel_stats.md | cut -c1-140
bash tools/profile_infer_step.sh > /dev/null 2>&1; head -4 gpurun_out/r05_b_inference_replay_kernel_timeline.md | cut -c1-140
bash tools/profile_train_step.sh > /dev/null 2>&1; head -3 gpurun_out/r05_a_train_replay_kernel_timeline.md
timeout 300 python tools/pointwise_bench.py 8 2>/dev/null > gpurun_out/r05_pointwise_layers.md; cat gpurun_out/r05_pointwise_layers.md
