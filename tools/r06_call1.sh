#!/bin/bash
# r06 call 1: same-box baseline + the existing LDS-patch kernel forced onto the wide DCN layers
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c1; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
for opt in "" "dcn_patch=8" "dcn_patch=6" "dcn_patch=5" "dcn_patch=3" "dcn_patch=2"; do
  echo "## opts: $opt" >> $O/dcn_layers.md
  timeout 300 python tools/dcn_layers_bench.py 8 3.0 "$opt" >> $O/dcn_layers.md 2>> $O/dcn_layers.err
done
for std in 1.5 5.0; do
  echo "## std $std default" >> $O/dcn_layers.md
  timeout 300 python tools/dcn_layers_bench.py 8 $std "" >> $O/dcn_layers.md 2>> $O/dcn_layers.err
  echo "## std $std dcn_patch=8" >> $O/dcn_layers.md
  timeout 300 python tools/dcn_layers_bench.py 8 $std "dcn_patch=8" >> $O/dcn_layers.md 2>> $O/dcn_layers.err
done
