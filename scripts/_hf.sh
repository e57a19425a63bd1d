cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TFRA_OWN_HF_DEBUG=1 timeout 600 python bench.py --config c3 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | grep "own_hf\|\[bench\]" > gpurun_out/s6_dbg.log
