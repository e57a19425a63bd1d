"""Prints the numbers of a bench.py JSON line that DESIGN.md / README.md quote.
  python scripts/show_bench_line.py profiles/r04_bench_line_driver_args.json ..."""
import json, sys
for f in sys.argv[1:]:
  d=json.loads(open(f).read().strip().splitlines()[-1])
  print('==', f, 'steps', d['steps'])
  print('value',round(d['value']/1e9,3),'us',round(d['ms_per_step']*1e3,2))
  for k in ('value_look_ahead_driver','value_plain_call','value_op_surface','value_op_surface_host_read_first','value_op_surface_find_first','value_accum','value_op_surface_table_ops_only'): print(' ',k, round(d[k]/1e9,3))
  r=d['roofline']; print(' frac',round(r['frac'],3),'step_frac',round(r['step_frac'],3),'traffic',r['traffic'], 'kernel us', round(r['avg_launch_us'],2))
  for n,x in r['kernels'].items(): print('   ', n[:50], round(x['avg_launch_us'],2), round(x.get('frac') or 0,3), x.get('traffic'))
  c=d['config']; print(' host', c['host_enqueue_ms_per_step'], c['overlapped_step_stats']['sequential'], 'U',c['unique_keys_per_batch'], 'ops/s', round(c['table_ops_per_s']/1e9,3), 'plan', round(c['plan_build_us_alone'],1), 'uniq', round(c['tfra_unique_us_alone'],1))
  cb=d['cpu_baseline']; print(' cpu', round(cb['value']/1e6,1), cb['resident_keys'], cb['cores'], cb['rungs_tried'])
  print(' per_op', cb['per_op'])
  sp=d.get('scaling_point'); print(' scaling_point', round(sp['value']/1e9,3), round(sp['ms_per_step']*1e3,1))
  for k,v in d['secondary'].items(): print(' ',k, round(v.get('value')/1e9,3), round((v.get('value_look_ahead_driver') or 0)/1e9,3), round(v.get('ms_per_step')*1e3,1), v.get('roofline',{}).get('traffic'), round(v.get('roofline',{}).get('step_frac') or 0,3))
  print(' verified', all(v is not False for v in c['verified'].values()))
