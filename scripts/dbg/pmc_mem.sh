#!/bin/bash
# memory-side counter passes over the overlapped step (and its roles alone): request counts, latencies, TLB, stalls
OUT=/root/repo/gpurun_out/pmc_mem
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for AB in 0 3 5; do
i=0
for C in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TCP_GATE_EN1_sum" \
         "TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_LEVEL_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
         "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCC_EA_WRREQ_LEVEL_sum TCC_EA_RDREQ_DRAM_CREDIT_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/a${AB}_p$i -o p -- python /root/repo/scripts/mb_overlap.py --skip-old --variants ${VARIANT:-2} --ablate $AB --depth 4 --steps 32 --out $OUT/mb.json > $OUT/log_a${AB}_$i.txt 2>&1
done
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_mem/*/p_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'step_k' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in acc.items():
        v = v[len(v)*5//8:]     # (the ablation starts 40 steps in)
        print(f.split('/')[-2], c, 'n=%d' % len(v), 'avg=%.1f' % (sum(v)/len(v)))
PY
