mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for shp in "8 96 320 64 64" "8 48 160 128 128" "8 24 80 256 256"; do
cd /tmp && rm -rf /tmp/p_x && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/p_x -o r -- python $R/tools/one_op.py dcnbwd $shp --reps 5 > /dev/null 2>&1
cd $R; DB=$(find /tmp/p_x -name "*.db" | head -1); echo "shape $shp"; python tools/prof_summary.py $DB | grep -E "dcn_bwd_tile" | cut -c1-110
done
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "dcn" 2>&1 | tail -2
