set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r2c_pytest.log 2>&1; echo pytest rc $?; tail -3 gpurun_out/r2c_pytest.log
# 2-rank rehearsal of the strong-scaling path on ONE device (gloo for the barrier; RCCL refuses two ranks per GPU)
XLZ_BENCH_DEVICE=0 XLZ_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r2c_bench_n2.json 2> gpurun_out/r2c_bench_n2.err; echo n2 rc $?; tail -4 gpurun_out/r2c_bench_n2.err; cat gpurun_out/r2c_bench_n2.json
bash tools/profile_bench.sh r02a cfg2-T > gpurun_out/r2c_prof_T.log 2>&1; echo profT rc $?; tail -25 gpurun_out/r2c_prof_T.log
