cd /root/repo
export TFRA_BENCH_DETAIL_DIR=/tmp
for V in "TFRA_APPLY_SPLIT=0" "TFRA_APPLY_SPLIT=1" "TFRA_APPLY_SPLIT=0" "TFRA_APPLY_SPLIT=1"; do
  env $V python bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -a '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$V', 'us/step %.1f host %.1f value %.3g' % (d['ms_per_step']*1e3, d['config']['host_enqueue_ms_per_step']*1e3, d['value']))"
  python - <<'PY'
import json
d=json.load(open('/tmp/bench_detail.json'))
for k,v in d['roofline']['kernels'].items(): print('   ', k[:70], v.get('avg_launch_us'))
PY
done
