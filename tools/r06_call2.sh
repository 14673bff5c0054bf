#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c2; mkdir -p $O
timeout 300 python tools/dcn_lds_check.py > $O/check.txt 2>&1
echo "check rc=$?" >> $O/check.txt
for opt in "dcn_lds=0" "dcn_lds=1" "dcn_lds=2"; do
  echo "## opts: $opt" >> $O/dcn_layers.md
  timeout 300 python tools/dcn_layers_bench.py 8 2.5 "$opt" >> $O/dcn_layers.md 2>> $O/dcn_layers.err
done
timeout 600 python -m pytest tests/test_gpu_bf16_kernels_vs_oracle.py tests/test_gpu_dcn_surface.py -x -q -m gpu > $O/pytest.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
