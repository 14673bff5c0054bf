mkdir -p gpurun_out
python -m pytest "tests/test_gpu_e2e.py::test_c5_batch32_hipgraph_fp32_rows_equal_reference_golden" -q -m gpu -x -p no:cacheprovider > gpurun_out/t2a.log 2>&1
python -m pytest "tests/test_gpu_train_step.py::test_ddp_over_rccl_wraps_the_hip_autograd_functions" -q -m gpu -x -p no:cacheprovider > gpurun_out/t2b.log 2>&1
python -m pytest "tests/test_gpu_train_step.py::test_graphed_train_step_equals_the_eager_step" -q -m gpu -x -p no:cacheprovider > gpurun_out/t2c.log 2>&1
python -m pytest tests/test_gpu_train_step.py -q -m gpu -p no:cacheprovider -k "sync_batchnorm or non_current or checkpoint" > gpurun_out/t2d.log 2>&1
tail -30 gpurun_out/t2a.log; tail -30 gpurun_out/t2b.log; tail -40 gpurun_out/t2c.log | cut -c1-300; tail -30 gpurun_out/t2d.log
