#!/bin/bash
# One-at-a-time sweep of the training step's workgroup-count / variant switches through MFX_OPTIONS (bench.py --mode train, one repeat of 30 steps each).
#   usage (GPU box): bash tools/train_option_sweep.sh [tag]   -> gpurun_out/<tag>_train_option_sweep.md
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_train_option_sweep.md
echo "| MFX_OPTIONS | ms per step | images/s |" > $OUT; echo "|---|---|---|" >> $OUT
for o in "" "wgrad_ws_blocks=600" "wgrad_ws_blocks=2400" "bn_blocks=512" "bn_blocks=1536" "bn_apply_blocks=512" "bn_apply_blocks=2048" "wgrad_patch_blocks=128" "wgrad_patch_blocks=512" \
         "wgrad_tr_blocks=256" "wgrad_tr_blocks=1024" "dcn_bt_fuse_blocks=128" "dcn_bt_fuse_blocks=256" "wgrad_mfma=3" "dcn_wgrad_m=1024" "wgrad_patch_waves=6" ""; do
  r=$(MFX_OPTIONS="$o" timeout 300 python bench.py --mode train --legs none --no-cpu-baseline --train-steps 30 --train-warmup 5 --train-repeats 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f | %.1f' % (d['ms_per_step'], d['value']))" 2>/dev/null)
  echo "| ${o:-(defaults)} | $r |" >> $OUT
done
cat $OUT
