cd /root/repo
timeout 300 python -m pytest tests/test_gpu_frontend.py -x -q -k "find_n" 2>&1 | tail -5
timeout 400 python bench.py --config m1b --no-secondary --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/s50.json 2> gpurun_out/s50.err; tail -3 gpurun_out/s50.err
