#!/bin/bash
# One-at-a-time sweep of the inference step's kernel-selection switches through MFX_OPTIONS (bench.py headline, B = 8 bf16, one graph).
#   usage (GPU box): bash tools/infer_option_sweep.sh [tag]   -> gpurun_out/<tag>_infer_option_sweep.md
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_infer_option_sweep.md
echo "| MFX_OPTIONS | ms per step | images/s |" > $OUT; echo "|---|---|---|" >> $OUT
for o in "" "dcn_wave=0" "dcn_wave=4" "dcn_wave=6" "dcn_ksplit=1" "dcn_tile=6" "dcn_tile=4" "kc=4" "halo_cw=2" "halo_cw=0" "cw_rows6=0" "dcn_fuse_off=0" "dcn_patch_fn8=1" "heads_planes=1" "cat_tile=4" "conv_tile=4" ""; do
  r=$(MFX_OPTIONS="$o" timeout 300 python bench.py --legs none --no-cpu-baseline --no-families 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f | %.1f' % (d['ms_per_step'], d['value']))" 2>/dev/null)
  echo "| ${o:-(defaults)} | $r |" >> $OUT
done
cat $OUT
