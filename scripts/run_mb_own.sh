#!/bin/bash
# GPU box: write-back timings on the 10^9-slot table: per-kernel (rocprofv3 --kernel-trace --stats) and the step drivers
cd /root/repo
rm -f gpurun_out/mb_own.jsonl
export TMPDIR=/tmp
( cd /tmp && MB_QUICK=1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_own -o own -- python /root/repo/scripts/mb_own.py ${SLOTS:-1000000000} prof > /root/repo/gpurun_out/mb_own_prof.log 2>&1 )
timeout 300 python scripts/mb_own.py ${SLOTS:-1000000000} full > gpurun_out/mb_own_full.log 2>&1 || { echo "full run failed"; tail -5 gpurun_out/mb_own_full.log; }
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/mb_own.jsonl'):
    r=json.loads(l)
    print(r['tag'], ' '.join('%s: U %d find %.1f planned %.1f direct %.1f%s' % (w, r[w]['U'], r[w]['find_us'], r[w]['upsert_planned_us'], r[w]['insert_unique_us'],
          (' | steps %.1f %.1f %.1f ok %s %s' % (r[w]['step_prefetch_us'], r[w]['step_plain_us'], r[w]['step_find_insert_unique_us'], r[w]['insert_unique_rows_ok'], r[w]['planned_rows_ok'])) if 'step_plain_us' in r[w] else '') for w in ('m1b','c3')))
PY
python - <<'PY'
import csv, glob
for f in glob.glob('/root/repo/gpurun_out/prof_own/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        n=row['Name']
        if any(k in n for k in ('upsert_','find_kernel','csr_')):
            print('%-60s calls %5s avg %9.1f us min %9.1f max %9.1f' % (n.replace('(anonymous namespace)::','')[:60], row['Calls'], float(row['AverageNs'])/1e3, float(row['MinNs'])/1e3, float(row['MaxNs'])/1e3))
    import shutil; shutil.copy(f, '/root/repo/gpurun_out/own_kernel_stats.csv')
PY
rm -rf gpurun_out/prof_own
