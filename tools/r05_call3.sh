mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
echo "== base"; MFX_LIB_PATH=$R/build_variants/lib_base.so python tools/dcn_layers_bench.py 8 3.0 | tee gpurun_out/r05_c3_dcn_base.md
echo "== new (om prefetch)"; python tools/dcn_layers_bench.py 8 3.0 | tee gpurun_out/r05_c3_dcn_new.md
timeout 300 python -m pytest tests/test_gpu_bf16_kernels_vs_oracle.py -q -x -p no:cacheprovider -k dcn 2>&1 | tail -2
