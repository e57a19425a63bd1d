cd /root/repo
export TFRA_BENCH_DETAIL_DIR=/tmp
for V in "X=0" "TFRA_APPLY_GRID_CAP=1024" "TFRA_APPLY_GRID_CAP=1100" "TFRA_APPLY_GRID_CAP=768" "TFRA_APPLY_GRID_CAP=512" "X=0"; do
  env $V python bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -a '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
dd=json.load(open('/tmp/bench_detail.json'))
k=[v.get('avg_launch_us') for kk,v in dd['roofline']['kernels'].items() if 'hot_sums' in kk][0]
print('$V', 'us/step %.1f host %.1f gradient-half-alone %.1f' % (d['ms_per_step']*1e3, d['config']['host_enqueue_ms_per_step']*1e3, k))"
done
