echo "== tr on (blocks 512)"; python tools/wgrad_bench.py --opts wgrad_tr_blocks=512
echo "== tr on (blocks 256)"; python tools/wgrad_bench.py --opts wgrad_tr_blocks=256
echo "== tr off"; python tools/wgrad_bench.py --opts wgrad_tr=0
