#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c4; mkdir -p $O
timeout 300 python tools/dcn_ps_check.py > $O/check.txt 2>&1
echo "check rc=$?" >> $O/check.txt
for opt in "dcn_ps=0" "dcn_ps=1"; do
  echo "## opts: $opt" >> $O/dcn_layers.md
  timeout 300 python tools/dcn_layers_bench.py 8 3.0 "$opt" >> $O/dcn_layers.md 2>> $O/dcn_layers.err
done
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o ps -- python $GRAFT_REPO_ROOT/tools/dcn_layers_bench.py 8 3.0 "dcn_ps=1" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $O/ps_kernel_stats.csv; rm -rf $O/prof
