# the round-end checks as the driver runs them, plus the round's artefacts: full `-m gpu` suite, smoke(), bench lines, training profile
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/full_gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|Error" gpurun_out/full_gpu_tests.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/round_artifacts.sh
