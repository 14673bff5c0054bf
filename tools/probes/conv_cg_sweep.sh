cd $GRAFT_REPO_ROOT
for cg in 0 32 64; do
echo "== halo_cg=$cg"
python tools/conv_bench.py 8 96 320 64 64 3,6,12 halo_cg=$cg
python tools/conv_bench.py 8 48 160 128 128 4,7,14,11 halo_cg=$cg
python tools/conv_bench.py 8 24 80 256 256 4,11,14 halo_cg=$cg
python tools/conv_bench.py 8 12 40 512 512 11,14,13 halo_cg=$cg
done
