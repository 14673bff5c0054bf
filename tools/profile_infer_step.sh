# rocprofv3 kernel trace of the inference step (single-stream form, so that the dispatch sequence is the network's order): per-kernel
# totals, idle gaps and the dispatch sequence of one graph replay.
# usage: bash tools/profile_infer_step.sh [dtype = bf16] [first kernel of a step = f1_fused] [output tag = b_inference]
DT=${1:-bf16}; MARK=${2:-f1_fused}; TAG=${3:-b_inference}
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/prof_inf1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf1 -o prof -- python $R/bench.py --dtype $DT --legs none --no-cpu-baseline --no-families --streams 1 --steps 20 --repeats 1 > $R/gpurun_out/prof_inf1.log 2>&1
cd $R; DB=$(find /tmp/prof_inf1 -name "*.db" | head -1)
python tools/gap_analysis.py $DB "$MARK" 5 gpurun_out/${RTAG:-r05}_${TAG}_step_sequence.txt > gpurun_out/${RTAG:-r05}_${TAG}_replay_kernel_timeline.md 2>&1; head -12 gpurun_out/${RTAG:-r05}_${TAG}_replay_kernel_timeline.md | cut -c1-130
