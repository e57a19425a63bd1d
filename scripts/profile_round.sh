#!/bin/bash
# Collects the rocprofv3 evidence for one round: kernel-trace stats + PMC passes (separately, as the
# MI355X guide prescribes).  Usage: bash scripts/profile_round.sh r01   (on the GPU box, via gpurun)
TAG=${1:-r01}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python /root/repo/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o $TAG -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_pmc_$N.log 2>&1
done
ls -R $OUT | head -40
