# kernel trace of the metric's step through the route driver at one rank (bench.py --config m1s --gpus 1): per-kernel durations
#   bash scripts/prof_aroute.sh [slots] [tag]
SLOTS=${1:-500000000}
TAG=${2:-r06}
cd /root/repo
export TMPDIR=/tmp TFRA_BENCH_DETAIL_DIR=/tmp
mkdir -p gpurun_out/prof_aroute
python bench.py --config m1s --gpus 1 --shard-slots $SLOTS --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/prof_aroute/plain.log 2> gpurun_out/prof_aroute/plain.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_aroute -o m1s -- python /root/repo/bench.py --config m1s --gpus 1 --shard-slots $SLOTS --steps 40 --warmup 10 --no-cpu-baseline > /root/repo/gpurun_out/prof_aroute/trace.log 2>&1)
find /tmp/prof_aroute -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_aroute/${TAG}_m1s_kernel_stats.csv
grep -a '^{"metric"' gpurun_out/prof_aroute/plain.log | tail -1 > gpurun_out/prof_aroute/${TAG}_m1s_bench_line.json
grep -a '^{"metric"' gpurun_out/prof_aroute/trace.log | tail -1 > gpurun_out/prof_aroute/${TAG}_m1s_bench_line_under_rocprof.json
head -25 gpurun_out/prof_aroute/${TAG}_m1s_kernel_stats.csv | cut -c1-220
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/prof_aroute/r06_m1s_bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step','verified')}, d['config'].get('host_enqueue_ms_per_step'), d['roofline'].get('avg_launch_us'), d['roofline'].get('step_frac'))
PY
