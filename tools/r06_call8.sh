#!/bin/bash
# same-box A/B of the inference step: old LDS-patch kernel vs dcn_lds (rows 16 / 8), project-then-sample on / off
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c8; mkdir -p $O
B="python bench.py --legs none --no-cpu-baseline --no-families --steps 30 --warmup 10 --repeats 5"
run() { echo "## $1" >> $O/ab.txt; env $2 $B --opts "$3" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['timing']['ms_per_step_each'])" >> $O/ab.txt; }
run "old patch kernel, ps off" "MFX_DCN_PS=0" "dcn_lds=0"
run "dcn_lds rows 16, ps off" "MFX_DCN_PS=0" "dcn_lds_rows=16"
run "dcn_lds rows 8, ps off" "MFX_DCN_PS=0" "dcn_lds_rows=8"
run "old patch kernel, ps on" "MFX_DCN_PS=1" "dcn_lds=0"
run "dcn_lds rows 16, ps on" "MFX_DCN_PS=1" "dcn_lds_rows=16"
run "old patch kernel, ps off (again)" "MFX_DCN_PS=0" "dcn_lds=0"
