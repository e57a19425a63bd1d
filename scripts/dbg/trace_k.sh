#!/bin/bash
OUT=/root/repo/gpurun_out/trace_k
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python /root/repo/scripts/mb_overlap.py --variants ${VARIANTS:-0,8} --depth 1 --steps 64 --skip-old --out $OUT/mb.json > $OUT/log.txt 2>&1
grep -E "step_k|setplan|find_kernel|upsert_own|upsert_rest" $OUT/t_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-200
grep -o "variant.: [0-9]*, .steps_per_host_call.: [0-9]*, .us_per_step.: [0-9.]*" $OUT/log.txt
