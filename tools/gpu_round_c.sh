mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py -q -m gpu -p no:cacheprovider -k "dcn" > gpurun_out/t3.log 2>&1
tail -25 gpurun_out/t3.log | cut -c1-300
python tools/train_layer_bench.py > gpurun_out/dcnbwd_big.json 2> gpurun_out/dcnbwd_big.err; cat gpurun_out/dcnbwd_big.json; tail -3 gpurun_out/dcnbwd_big.err
for s in "8 48 160 128 64" "8 24 80 256 128" "8 12 40 512 256"; do python tools/one_op.py dcnbwd $s --reps 5 2>&1 | tail -1; done
