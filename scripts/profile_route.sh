#!/bin/bash
# Evidence for the multi-GPU step on ONE GPU with the collectives forced on (RCCL world of 1): step times of the three
# drivers + rocprofv3 kernel stats of the native one.   bash scripts/profile_route.sh r02      (on the GPU box, via gpurun)
TAG=${1:-r02}
OUT=/root/repo/gpurun_out/route_$TAG
mkdir -p $OUT
cd /root/repo
bash scripts/run_forced_route.sh 200 > $OUT/forced_route_steps.txt 2>&1
TFRA_ROUTE_TIMING=1 python scripts/profile_route_native.py rccl 2>&1 | grep "us per step\|tfra_route" > $OUT/native_host_stages.txt
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29778 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
cd /tmp && export TMPDIR=/tmp
TFRA_BENCH_FORCE_A2A=1 TFRA_BENCH_ROUTE=native rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- \
  python /root/repo/bench.py --config c2 --no-secondary --no-cpu-baseline --steps 200 --warmup 20 > $OUT/bench_trace.log 2>&1
cat $OUT/forced_route_steps.txt
