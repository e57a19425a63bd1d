#!/bin/bash
# PMC passes over the overlapped step (FETCH_SIZE / WRITE_SIZE / L2 hit), per-kernel averages printed
OUT=/root/repo/gpurun_out/pmc_step
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-12)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -o p -- python /root/repo/scripts/mb_overlap.py --variants ${VARIANT:-0} --depth 1 --steps 32 --out $OUT/mb_$N.json > $OUT/log_$N.txt 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_step/*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for key in ('step_k', 'step_rest', 'find_kernel', 'upsert_own', 'upsert_rest', 'setplan'):
            if key in k:
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key, d in acc.items():
        for c, v in d.items():
            v = v[len(v)//4:]
            print(f.split('/')[-2], key, c, 'n=%d' % len(v), 'avg=%.3f' % (sum(v)/len(v)))
PY
