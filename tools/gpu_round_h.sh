python tools/probes/graph_memset_probe.py 2>&1 | grep "replay \|eager"
python tools/probes/graph_alloc_probe.py fp32 2>&1 | grep "consistent"
python -m pytest tests/test_gpu_train_step.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
python bench.py --mode train --no-cpu-baseline --steps 5 2>&1 | tail -1 | cut -c90-200
