#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c15; mkdir -p $O
B="python bench.py --legs none --no-cpu-baseline --no-families --steps 30 --warmup 10 --repeats 5"
run() { echo "## $1" >> $O/ab.txt; env $2 $B --opts "$3" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'])" >> $O/ab.txt; }
run "default (dcn_lds auto, ps <= 24 MB)" "X=1" ""
run "dcn_lds=2 (also 128->64 and 256->64)" "X=1" "dcn_lds=2"
run "dcn_lds=2, ps off" "MFX_DCN_PS=0" "dcn_lds=2"
run "dcn_lds=0 ps off (r05 kernels)" "MFX_DCN_PS=0" "dcn_lds=0"
run "default again" "X=1" ""
