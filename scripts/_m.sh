cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TFRA_OWN_HF=1 timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > gpurun_out/s13_tests_hf1.log
timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/s13_tests_auto.log
for hf in 0 auto 0 auto; do
  if [ $hf = auto ]; then unset TFRA_OWN_HF; else export TFRA_OWN_HF=$hf; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > /dev/null
  python - <<PY >> gpurun_out/s13_ab.log
import json
d=json.load(open('bench_detail.json'))
ks=d['roofline']['kernels']
print('$hf', {k[12:]:round(v*1e3,2) for k,v in d.items() if k.startswith('ms_per_step_') and ('op_surface' in k or 'accum' in k or 'look' in k or 'plain' in k)}, [round(v['avg_launch_us'],2) for k,v in ks.items() if 'upsert_own' in k], all(v for v in d['config']['verified'].values() if isinstance(v,bool)))
PY
done
