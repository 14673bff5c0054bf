#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c11; mkdir -p $O
timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider -s -k "training_step_vs_oracle or dcn_lds or project_then_sample or test_conv2d or gram_heads or heads_fused" > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error|full-size|vs reference" $O/gpu_tests.log | tail -40 > $O/gpu_tests_tail.txt
for o in "heads_mfma32=0" "heads_mfma32=1" "heads_mfma32=0" "heads_mfma32=1"; do
  python tools/one_op.py heads --batch 8 --dtype bf16 --reps 6 --opts "$o" 2>/dev/null | tail -1 >> $O/heads_ab.txt
done
python tools/one_op.py heads --batch 8 --dtype fp16 --reps 6 --opts "heads_mfma32=0" 2>/dev/null | tail -1 >> $O/heads_ab.txt
python tools/one_op.py heads --batch 8 --dtype fp16 --reps 6 --opts "heads_mfma32=1" 2>/dev/null | tail -1 >> $O/heads_ab.txt
python tools/one_op.py heads --batch 32 --dtype bf16 --reps 3 --opts "heads_mfma32=0" 2>/dev/null | tail -1 >> $O/heads_ab.txt
python tools/one_op.py heads --batch 32 --dtype bf16 --reps 3 --opts "heads_mfma32=1" 2>/dev/null | tail -1 >> $O/heads_ab.txt
