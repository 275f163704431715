timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or host_pipeline or many_streams or baseline_shape" > gpurun_out/r2g_pytest.log 2>&1; echo pytest rc $?; tail -5 gpurun_out/r2g_pytest.log
XLZ_DEBUG=1 timeout -k 10 400 python tools/host_path.py T 16384 1048576 1024 2>&1 | tail -7
XLZ_DEBUG=1 timeout -k 10 400 python tools/host_path.py T 65536 65536 8192 2>&1 | tail -4
XLZ_DEBUG=nopipe timeout -k 10 400 python tools/host_path.py T 65536 65536 8192 2>&1 | tail -4
