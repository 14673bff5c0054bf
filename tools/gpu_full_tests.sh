mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/full_gpu_tests.log 2>&1; tail -15 gpurun_out/full_gpu_tests.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
