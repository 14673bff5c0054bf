mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
for o in "dcn_wave=1" "dcn_wave=4" "dcn_wave=5" "dcn_wave=6" "dcn_wave=8" "dcn_tile=6" "dcn_tile=5" "dcn_tile=3"; do echo "== $o"; python tools/dcn_layers_bench.py 8 3.0 $o 2>&1 | grep -v amdgpu | cut -c1-80; done
