# the one-rank route (bench.py --config m1s --gpus 1) under a few settings: step time, host time, owner launch
cd /root/repo
export TFRA_BENCH_DETAIL_DIR=/tmp
for V in "TFRA_RP_IPT=1" "TFRA_RP_IPT=2" "TFRA_RP_IPT=4" "TFRA_ROUTE_AHEAD=3 TFRA_RP_IPT=4"; do
  env $V python bench.py --config m1s --gpus 1 --shard-slots ${1:-500000000} --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -a '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$V', 'us/step %.1f host %.1f owner_launch %.1f verified %s' % (d['ms_per_step']*1e3, d['config']['host_enqueue_ms_per_step']*1e3, d['roofline']['avg_launch_us'], d['verified']))"
done
