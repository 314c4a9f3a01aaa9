cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_jackknife.py tests/test_gpu_mapper.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/r4_t11_tests.log 2>&1
tail -15 gpurun_out/r4_t11_tests.log
