# bench lines (the exact default command first) and the training-step profile + replay timeline of the current HEAD -> gpurun_out/r02_*
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python bench.py > gpurun_out/r02_final_bench_infer.json 2> gpurun_out/bench_infer.err; tail -c 600 gpurun_out/r02_final_bench_infer.json; echo
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline > gpurun_out/r02_final_bench_infer_fp32.json 2>/dev/null; cut -c1-260 gpurun_out/r02_final_bench_infer_fp32.json
timeout 600 python bench.py --batch 32 --no-cpu-baseline > gpurun_out/r02_final_bench_infer_b32.json 2>/dev/null; cut -c1-260 gpurun_out/r02_final_bench_infer_b32.json
timeout 900 python bench.py --mode train > gpurun_out/r02_final_bench_train.json 2> gpurun_out/bench_train.err; cut -c1-330 gpurun_out/r02_final_bench_train.json; tail -c 700 gpurun_out/r02_final_bench_train.json; echo
timeout 600 python bench.py --mode train --dtype fp32 --no-cpu-baseline > gpurun_out/r02_final_bench_train_fp32.json 2>/dev/null; cut -c1-260 gpurun_out/r02_final_bench_train_fp32.json
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r02 -- python $R/bench.py --mode train --no-cpu-baseline --steps 10 > $R/gpurun_out/prof_tr.log 2>&1
cd $R; DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools/prof_summary.py $DB > gpurun_out/r02_h_train_step_kernel_stats.md
python tools/gap_analysis.py $DB "mfx::bf16_t, mfx::bf16_t, 256, 16" 5 gpurun_out/r02_h_train_step_sequence.txt > gpurun_out/r02_h_train_replay_kernel_timeline.md 2>&1; head -12 gpurun_out/r02_h_train_replay_kernel_timeline.md | cut -c1-120
