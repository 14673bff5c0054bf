#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c7; mkdir -p $O
timeout 300 python tools/dcn_lds_check.py > $O/check.txt 2>&1
echo "check rc=$?" >> $O/check.txt
for opt in "dcn_lds=0" "dcn_lds_rows=16" "dcn_lds_rows=8"; do
  echo "## opts: $opt" >> $O/dcn_layers.md
  timeout 300 python tools/dcn_layers_bench.py 8 2.5 "$opt" >> $O/dcn_layers.md 2>> $O/dcn_layers.err
done
DCN_LDS_ROWS=8 MFX_LIB_PATH=build_variants/lib_probe.so timeout 300 python tools/probes/dcn_lds_probe.py module > $O/probe_module_r8.txt 2>&1
DCN_LDS_ROWS=16 MFX_LIB_PATH=build_variants/lib_probe.so timeout 300 python tools/probes/dcn_lds_probe.py module > $O/probe_module_r16.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
