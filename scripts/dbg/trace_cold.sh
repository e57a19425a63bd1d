#!/bin/bash
OUT=/root/repo/gpurun_out/trace_cold
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python /root/repo/scripts/mb_overlap.py --variants 0 --depth 4 --steps 20 --skip-old --out $OUT/mb.json > $OUT/log.txt 2>&1
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('/root/repo/gpurun_out/trace_cold/t_kernel_trace.csv')) if 'step_k' in r['Kernel_Name'] or 'step_rest' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
out=[]
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.append(('K' if 'step_k' in r['Kernel_Name'] else 'R', (e-s)/1000.0, (s-prev)/1000.0 if prev else 0.0))
    prev=e
for i in range(0,min(len(out),240),2):
    a=out[i]; b=out[i+1] if i+1<len(out) else ('',0,0)
    if i%8==0 or i<40: print(i//2, a[0], 'dur %.1f gap %.1f |'%(a[1],a[2]), b[0], 'dur %.1f gap %.1f'%(b[1],b[2]))
PY
