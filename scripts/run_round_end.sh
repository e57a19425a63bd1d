# What the driver runs at round end, in one gpurun call: the GPU test suite and the bench line with the driver's arguments
# (add `default` as the first argument for the 200 / 20 line too).
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu.log 2>&1; tail -2 gpurun_out/r04_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_args.json 2> gpurun_out/r04_bench_driver_args.err; tail -c 200 gpurun_out/r04_bench_driver_args.err
if [ "$1" = "default" ]; then
  timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; tail -c 200 gpurun_out/r04_bench_default.err
fi
