timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or host_pipeline or many_streams or baseline_shape" > gpurun_out/r2g_pytest.log 2>&1; echo pytest rc $?; tail -15 gpurun_out/r2g_pytest.log
timeout -k 10 300 python tools/host_path.py T 4096 1048576 1024 2>&1 | tee gpurun_out/r2g_host_path.log
