# the metric's step through the route driver at one rank with the REAL transport forced on (one-rank RCCL communicators: every alltoall a
# grouped ncclSend / ncclRecv to itself): bench line + kernel trace      bash scripts/prof_aroute_rccl.sh [slots] [tag]
SLOTS=${1:-500000000}
TAG=${2:-r06}
cd /root/repo
export TMPDIR=/tmp TFRA_BENCH_DETAIL_DIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=29777 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 TFRA_BENCH_FORCE_A2A=1
mkdir -p gpurun_out/prof_aroute
ARGS="--config m1s --gpus 1 --shard-slots $SLOTS --steps 40 --warmup 10 --no-cpu-baseline"
python bench.py $ARGS > gpurun_out/prof_aroute/rccl_plain.log 2> gpurun_out/prof_aroute/rccl_plain.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_aroute_rccl -o m1s -- python /root/repo/bench.py $ARGS > /root/repo/gpurun_out/prof_aroute/rccl_trace.log 2>&1)
find /tmp/prof_aroute_rccl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_aroute/${TAG}_m1s_rccl1_kernel_stats.csv
grep -a '^{"metric"' gpurun_out/prof_aroute/rccl_plain.log | tail -1 > gpurun_out/prof_aroute/${TAG}_m1s_rccl1_bench_line.json
grep -i "step_k\|gather_rows\|routeplan\|rccl\|copyBuffer" gpurun_out/prof_aroute/${TAG}_m1s_rccl1_kernel_stats.csv | cut -c1-200
python - <<PY
import json
d=json.load(open('/root/repo/gpurun_out/prof_aroute/${TAG}_m1s_rccl1_bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step','verified')}, d['config'].get('host_enqueue_ms_per_step'), d['config'].get('rccl_ranks_seen'), d['config'].get('parallelism'))
PY
tail -3 gpurun_out/prof_aroute/rccl_plain.err
