#!/bin/bash
# SQ counter passes over the overlapped step (scripts/mb_overlap.py): instructions, busy and wait cycles per launch of step_k.  TCP_* / TA_* counters abort on this image (and hang): do not add them.
OUT=/root/repo/gpurun_out/pmc_step_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o p -- python /root/repo/scripts/mb_overlap.py --skip-old --variants ${VARIANT:-2} --ablate ${ABLATE:-0} --depth 4 --steps 32 --out $OUT/mb_$i.json > $OUT/log_$i.txt 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_step_sq/*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for key in ('step_k',):
            if key in k:
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key, d in acc.items():
        for c, v in d.items():
            v = v[len(v)//2:]
            print(f.split('/')[-2], key, c, 'n=%d' % len(v), 'avg=%.1f' % (sum(v)/len(v)))
PY
