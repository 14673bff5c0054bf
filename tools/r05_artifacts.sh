# round-5 artefacts of the current HEAD -> gpurun_out/r05_* (copied into profiles/ by hand): bench lines, kernel statistics and replay timelines of the
# inference and training steps, PMC passes (heads, DCN modules, trunk 3x3, DCN backward), A/B tables of the new kernels
export RTAG=r05; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
bash tools/round_artifacts.sh
python tools/conv_cw_bench.py 8 > gpurun_out/r05_conv_cw_ab.md 2>/dev/null; cat gpurun_out/r05_conv_cw_ab.md
python tools/dcn_layers_bench.py 8 3.0 2>/dev/null > gpurun_out/r05_dcn_layers.md; cat gpurun_out/r05_dcn_layers.md
bash tools/pmc_dcn.sh r05 > /dev/null 2>&1; head -40 gpurun_out/r05_dcn_pmc.txt
bash tools/pmc_conv.sh r05 > /dev/null 2>&1; head -30 gpurun_out/r05_conv_pmc.txt
