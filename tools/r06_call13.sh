#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c13; mkdir -p $O
for o in "heads_mfma32=0" "heads_mfma32=1" "heads_mfma32=0" "heads_mfma32=1" "heads_mfma32=0" "heads_mfma32=1"; do
  python tools/one_op.py heads --batch 8 --dtype bf16 --reps 6 --opts "$o" 2>/dev/null | tail -1 >> $O/heads_ab.txt
done
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "heads_fused" > $O/t.log 2>&1; tail -2 $O/t.log >> $O/heads_ab.txt
