#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
for R in 0.5 0.0; do
  D=/root/repo/gpurun_out/prof_sweep_$R
  rm -rf $D
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $D -o x -- python /root/repo/scripts/mb_sweep.py 1000000000 $R > /root/repo/gpurun_out/sweep_$R.log 2>&1 ) || echo FAILED $R
  tail -c 600 /root/repo/gpurun_out/sweep_$R.log
  python - $D <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'upsert_own_kernel' in r['Kernel_Name'] or 'upsert_rest_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    own = [r for r in rows if 'upsert_own' in r['Kernel_Name']]
    fin = [r for r in rows if 'upsert_rest' in r['Kernel_Name']]
    by = collections.OrderedDict()
    for o, fi in zip(own, fin):
        g = int(o['Grid_Size_X']) if 'Grid_Size_X' in o else int(o.get('Grid_Size', 0))
        by.setdefault(g, []).append(((int(o['End_Timestamp']) - int(o['Start_Timestamp'])) / 1e3, (int(fi['End_Timestamp']) - int(fi['Start_Timestamp'])) / 1e3))
    for g, v in by.items():
        print('grid %7d threads (~%6d keys): own %s | rest %s' % (g, g // 4, ' '.join('%.1f' % a for a, _ in v), ' '.join('%.1f' % b for _, b in v)))
PY
  rm -rf $D
done
