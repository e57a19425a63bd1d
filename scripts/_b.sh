cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/s7_line.json 2> gpurun_out/s7_err.log ) 2> gpurun_out/s7_time.log
cp bench_detail.json gpurun_out/s7_detail.json
