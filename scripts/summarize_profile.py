"""Turns the rocprofv3 output of scripts/profile_round.sh (gpurun_out/prof_<tag>/) into the tracked
evidence files profiles/<tag>_kernel_stats.csv and profiles/<tag>_summary.json.

  python scripts/summarize_profile.py r01

HBM traffic per launch = FETCH_SIZE + WRITE_SIZE from the separate --pmc passes, corrected as
MI355X_MICROARCH.md prescribes (values are KiB; gfx950 reports half of the wide coalesced reads, so
FETCH_SIZE is doubled)."""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["insert_unique_kernel", "bucket_merge_kernel", "tile_reduce_kernel", "find_kernel", "apply_kernel",
           "tile_reduce_kernel<plan>", "bucket_merge_kernel<plan>", "tile_sums_kernel", "bucket_sums_kernel"]


def short(name):
  m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
  if not m:
    return None
  k = m.group(1)
  if k in ("tile_reduce_kernel", "bucket_merge_kernel") and m.group(2) and "true" in m.group(2):
    k += "<plan>"   # id-only half (tfra_sparse_plan_build)
  return k


def main():
  tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
  src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
  out_dir = os.path.join(ROOT, "profiles")
  os.makedirs(out_dir, exist_ok=True)
  shutil.copy(os.path.join(src, "trace", tag + "_kernel_stats.csv"), os.path.join(out_dir, tag + "_kernel_stats.csv"))
  kernels = {}
  with open(os.path.join(src, "trace", tag + "_kernel_stats.csv")) as f:
    for row in csv.DictReader(f):
      k = short(row["Name"])
      if k in KERNELS:
        d = kernels.setdefault(k, {"calls": 0, "total_ns": 0, "min_us": 1e30, "max_us": 0})
        d["calls"] += int(row["Calls"])
        d["total_ns"] += int(row["TotalDurationNs"])
        d["min_us"] = min(d["min_us"], int(row["MinNs"]) / 1e3)
        d["max_us"] = max(d["max_us"], int(row["MaxNs"]) / 1e3)
  for d in kernels.values():
    d["avg_us"] = round(d.pop("total_ns") / d["calls"] / 1e3, 2)
    d["min_us"], d["max_us"] = round(d["min_us"], 2), round(d["max_us"], 2)

  def counter_avg(sub, names):
    acc = {}
    path = os.path.join(src, sub, tag + "_counter_collection.csv")
    if not os.path.exists(path):
      return acc
    with open(path) as f:
      for row in csv.DictReader(f):
        k = short(row["Kernel_Name"])
        if k in KERNELS and row["Counter_Name"] in names:
          a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, set()])
          a[0] += float(row["Counter_Value"])
          a[1].add(row["Dispatch_Id"])
    return {k: {c: v[0] / max(len(v[1]), 1) for c, v in d.items()} for k, d in acc.items()}

  fetch = counter_avg("pmc_FETCH_SIZE", {"FETCH_SIZE"})
  write = counter_avg("pmc_WRITE_SIZE", {"WRITE_SIZE"})
  tcc = counter_avg("pmc_TCC_HIT_sum_TCC_MISS_sum", {"TCC_HIT_sum", "TCC_MISS_sum"})
  for k, d in kernels.items():
    f = fetch.get(k, {}).get("FETCH_SIZE")
    w = write.get(k, {}).get("WRITE_SIZE")
    if f is not None and w is not None:
      d["FETCH_SIZE_KiB_raw"] = round(f, 1)
      d["WRITE_SIZE_KiB"] = round(w, 1)
      d["hbm_bytes_per_launch_corrected"] = int((2 * f + w) * 1024)
    t = tcc.get(k)
    if t and (t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0)) > 0:
      d["l2_hit_rate"] = round(t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), 3)
  summary = {
      "round": int(re.sub(r"\D", "", tag) or 0),
      "command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline  [default "
                 "--plan prefetch: tile_sums/bucket_sums/apply on the main stream, <plan> kernels on the second stream; its "
                 "secondary fused-call loop contributes tile_reduce/bucket_merge]  (+ separate "
                 "--pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum passes, --steps 20)  [scripts/profile_round.sh, "
                 "scripts/summarize_profile.py]",
      "workload": "BASELINE configs[1]: 100M keys, dim 64 fp32 [p|m|v], Zipf-1.2 batch 131072",
      "kernels": kernels,
      "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
              "coalesced reads). apply_kernel mixes the INDIRECT launches of the step with the direct ones bench.py times "
              "separately.",
  }
  with open(os.path.join(out_dir, tag + "_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
  print(json.dumps({k: (v["avg_us"], v.get("hbm_bytes_per_launch_corrected")) for k, v in kernels.items()}))


if __name__ == "__main__":
  main()
