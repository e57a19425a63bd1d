#!/bin/bash
# Collects the rocprofv3 evidence for one round: per workload (m1b = the default bench line, c3 / c2 / c4 = its secondaries) a
# kernel-trace + stats run and separate PMC passes (as MI355X_MICROARCH.md prescribes; gpurun refuses --pmc combined
# with API tracing).   Usage (on the GPU box, via gpurun):  bash scripts/profile_round.sh r04 ["m1b:full m1s:io c3:io c2:io c4:io"]
#   <workload>:full = FETCH_SIZE, WRITE_SIZE, L2 hit / miss and the SQ instruction counters; :io = the two traffic passes only
TAG=${1:-r04}
WORKLOADS=${2:-"m1b:full m1s:io c3:io c2:io c4:io"}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp TFRA_BENCH_SKIP_ROUTED_LOCAL=1   # (m1b's trace holds the headline's step launches only; m1s = the routed step at one rank, its own workload)
for WS in $WORKLOADS; do
  W=${WS%%:*}; SET=${WS##*:}
  ARGS="--config $W --no-secondary --no-cpu-baseline"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_trace -o $TAG -- python /root/repo/bench.py $ARGS --steps 40 --warmup 10 > $OUT/${W}_bench_trace.log 2>&1
  CS=("FETCH_SIZE" "WRITE_SIZE")
  if [ "$SET" = "full" ]; then CS+=("TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"); fi
  for C in "${CS[@]}"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-24)
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${W}_pmc_$N -o $TAG -- python /root/repo/bench.py $ARGS --steps 10 --warmup 5 > $OUT/${W}_bench_pmc_$N.log 2>&1
  done
done
ls $OUT
