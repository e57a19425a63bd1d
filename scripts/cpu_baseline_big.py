"""The CPU baseline leg of bench.py alone, at a table size the default run's time box does not allow (the largest N the
host's RAM holds): `TFRA_BENCH_CPU_KEYS=1000000000 TFRA_BENCH_CPU_INIT_DIV=1 TFRA_BENCH_CPU_BOX_S=600 python scripts/cpu_baseline_big.py`
-> one JSON object (profiles/r05_cpu_baseline_1e9.json is its output on the round's GPU box host: 256 cores, 3.2 TB)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
  r = bench.cpu_baseline(131072)
  r.pop("per_op_per_core", None)
  print(json.dumps(r))
