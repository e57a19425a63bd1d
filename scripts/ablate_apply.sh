#!/bin/bash
# kernel-time ablation of tile_reduce (A) / bucket_merge (C): run under rocprofv3 with early exits.
# Needs a tuning build:  TFRA_WITH_TUNING=1 python recommenders-addons_amd/build.py --force   (and --plan off)
cd /tmp && export TMPDIR=/tmp
for v in "A 0 C 0" "A 1 C 0" "A 2 C 0" "A 3 C 0" "A 4 C 0" "A 0 C 1" "A 0 C 2" "A 0 C 3"; do
  set -- $v
  rm -rf /tmp/abl
  TFRA_DBG_STOP_A=$2 TFRA_DBG_STOP_C=$4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o x -- python /root/repo/bench.py --steps 30 --warmup 5 --keys 10000000 --no-cpu-baseline --plan off > /dev/null 2>&1
  python - <<PY
import csv
rows={r["Name"]:r for r in csv.DictReader(open("/tmp/abl/x_kernel_stats.csv"))}
def g(sub):
    for k,r in rows.items():
        if sub in k: return float(r["AverageNs"])/1e3
    return -1
print("stopA=$2 stopC=$4  tile_reduce=%.1fus bucket_merge=%.1fus apply=%.1fus find=%.1fus" % (g("tile_reduce"), g("bucket_merge"), g("apply_kernel"), g("find_kernel")))
PY
done
