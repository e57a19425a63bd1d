#!/bin/bash
# The N>1 step of bench.py (config c2) on ONE GPU with the collectives forced on (RCCL world of 1): native C driver,
# the Python-driven prefetched route, and the plain single-GPU step for reference.   bash scripts/run_forced_route.sh [steps]
K=${1:-200}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29777 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for R in native prefetch; do
  TFRA_BENCH_FORCE_A2A=1 TFRA_BENCH_ROUTE=$R python bench.py --config c2 --no-secondary --no-cpu-baseline --steps $K --warmup 20 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$R', 'ms_per_step', round(d['ms_per_step'], 4), 'host_enqueue', d['config'].get('host_enqueue_ms_per_step'), 'value', round(d['value'] / 1e9, 3), 'G pairs/s')"
done
python bench.py --config c2 --no-secondary --no-cpu-baseline --steps $K --warmup 20 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('single', 'ms_per_step', round(d['ms_per_step'], 4), 'value', round(d['value'] / 1e9, 3), 'G pairs/s')"
