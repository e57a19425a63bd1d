#!/bin/bash
# GPU box: the write-back variants, one process each (the knobs are read once per process)
cd /root/repo
rm -f gpurun_out/mb_own.jsonl
for V in "inside 512" "inside 256" "inside 1024" "kernel 512" "rest 512"; do
  set -- $V
  TFRA_OWN_FINISH=$1 TFRA_OWN_NT=$2 timeout 300 python scripts/mb_own.py ${SLOTS:-1000000000} "$1-$2" > gpurun_out/mb_own_$1_$2.log 2>&1 || echo "variant $V failed rc=$?"
  tail -c 300 gpurun_out/mb_own_$1_$2.log | tail -n 2
done
