"""Prints the runs of scripts/mb_overlap.py output files: variant, ablation mask, steps per host call, us per step, windows, role time stamps.
  python scripts/mb_overlap_show.py gpurun_out/mb_overlap.json ..."""
import json, sys
for f in sys.argv[1:]:
  d=json.load(open(f))
  for r in d['runs']:
    st=r.get('stats',{})
    print(r['variant'], r.get('ablate',0), r['steps_per_host_call'], round(r['us_per_step'],2), [round(x,1) for x in r['windows']], r['last_batch_ok'], 'seq',st.get('sequential'), 'def',st.get('deferred_evictions'), 'vic',st.get('victims_noted'))
    if r.get('timing'):
      for k,v in r['timing']["role_spans_us_median (start, end since the launch's first block)"].items(): print('   ',k,[round(x,1) for x in v])
