"""Turns the rocprofv3 output of scripts/profile_round.sh (gpurun_out/prof_<tag>/) into the tracked evidence files
profiles/<tag>_<workload>_kernel_stats.csv and profiles/<tag>_summary.json.

  python scripts/summarize_profile.py r02

HBM traffic per launch = FETCH_SIZE + WRITE_SIZE from the separate --pmc passes, corrected as MI355X_MICROARCH.md
prescribes (values are KiB; gfx950 reports half of the wide coalesced reads, so FETCH_SIZE is doubled).  The SQ pass
gives instructions and wave cycles per launch (quad-cycle units): what shows that the write-back kernels are bound by
instruction issue and dependent round trips, not by bytes."""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["find_kernel", "upsert_own_kernel", "upsert_rest_kernel", "setplan_kernel", "insert_unique_kernel",
           "insert_evict_kernel", "export_kernel", "csr_tile_kernel", "csr_bucket_kernel", "csr_scatter_kernel", "hot_sums_kernel",
           "apply_csr_kernel", "apply_kernel", "density_kernel", "unique_idx_kernel", "unq_insert_kernel", "gather_csr_kernel", "plan_dest_bins_kernel",
           "plan_dest_keys_kernel", "part_scatter_kernel", "accum_kernel", "step_k", "routeplan_insert_kernel", "routeplan_emit_kernel",
           "gather_rows16_kernel", "move_rows_kernel"]
WORKLOADS = ("m1b", "m1s", "c3", "c2", "c4")
COMMANDS = {w: "python bench.py --config %s --no-secondary --no-cpu-baseline" % w for w in WORKLOADS}
SRC_NAMES = {"0": "plan", "1": "direct", "2": "set"}   # upsert_own_kernel<G, SIMPLE, SRC> / upsert_rest_kernel<G, SRC>: where the keys come from


def short(name):
  if re.search(r"\bstep_k_u\d", name):   # the overlapped step's one launch (csrc/tfra_step_impl.h: step_k_u2 / step_k_u1 / step_k_u2_t)
    return "step_k"
  m = re.search(r"(\w+_kernel)\b", name)
  if not m or m.group(1) not in KERNELS:
    return None
  k = m.group(1)
  if k in ("upsert_own_kernel", "upsert_rest_kernel"):
    a = re.search(r"%s<([^>]*)>" % k, name)
    if a:
      args = [x.strip() for x in a.group(1).split(",")]
      src = args[2] if k == "upsert_own_kernel" and len(args) > 2 else args[1] if len(args) > 1 else "?"   # <G, SIMPLE, SRC, U> / <G, SRC>
      k += "[%s]" % SRC_NAMES.get(src, "?")
  return k


def counter_avg(path, names):
  acc = {}
  if not os.path.exists(path):
    return acc
  with open(path) as f:
    for row in csv.DictReader(f):
      k = short(row["Kernel_Name"])
      if k and row["Counter_Name"] in names:
        a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, set()])
        a[0] += float(row["Counter_Value"])
        a[1].add(row["Dispatch_Id"])
  return {k: {c: v[0] / max(len(v[1]), 1) for c, v in d.items()} for k, d in acc.items()}


def workload(src, tag, w, out_dir):
  stats = os.path.join(src, w + "_trace", tag + "_kernel_stats.csv")
  if not os.path.exists(stats):
    return None
  shutil.copy(stats, os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (tag, w)))
  kernels = {}
  with open(stats) as f:
    for row in csv.DictReader(f):
      k = short(row["Name"])
      if k:
        d = kernels.setdefault(k, {"calls": 0, "total_ns": 0, "min_us": 1e30, "max_us": 0})
        d["calls"] += int(row["Calls"])
        d["total_ns"] += int(row["TotalDurationNs"])
        d["min_us"] = min(d["min_us"], int(row["MinNs"]) / 1e3)
        d["max_us"] = max(d["max_us"], int(row["MaxNs"]) / 1e3)
  for d in kernels.values():
    d["avg_us"] = round(d.pop("total_ns") / d["calls"] / 1e3, 2)
    d["min_us"], d["max_us"] = round(d["min_us"], 2), round(d["max_us"], 2)
  cc = lambda sub: os.path.join(src, "%s_pmc_%s" % (w, sub), tag + "_counter_collection.csv")
  fetch = counter_avg(cc("FETCH_SIZE"), {"FETCH_SIZE"})
  write = counter_avg(cc("WRITE_SIZE"), {"WRITE_SIZE"})
  tcc = counter_avg(cc("TCC_HIT_sum_TCC_MISS_sum"), {"TCC_HIT_sum", "TCC_MISS_sum"})
  sqn = {"SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM"}
  sq = counter_avg(cc("SQ_WAVES_SQ_INSTS_VALU_S"), sqn)
  for k, d in kernels.items():
    f, wr = fetch.get(k, {}).get("FETCH_SIZE"), write.get(k, {}).get("WRITE_SIZE")
    if f is not None and wr is not None:
      d["FETCH_SIZE_KiB_raw"], d["WRITE_SIZE_KiB"] = round(f, 1), round(wr, 1)
      d["hbm_bytes_per_launch_corrected"] = int((2 * f + wr) * 1024)
    t = tcc.get(k)
    if t and (t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0)) > 0:
      d["l2_hit_rate"] = round(t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), 3)
    s = sq.get(k)
    if s and s.get("SQ_WAVES"):
      wv = s["SQ_WAVES"]
      d["sq_per_wave"] = {"waves_per_launch": round(wv), "valu": round(s.get("SQ_INSTS_VALU", 0) / wv), "salu": round(s.get("SQ_INSTS_SALU", 0) / wv),
                          "vmem": round(s.get("SQ_INSTS_VMEM", 0) / wv, 1), "wave_quad_cycles": round(s.get("SQ_WAVE_CYCLES", 0) / wv),
                          "wait_any_frac": round(s.get("SQ_WAIT_ANY", 0) / max(s.get("SQ_WAVE_CYCLES", 1), 1), 3),
                          "issue_stall_frac": round(s.get("SQ_WAIT_INST_ANY", 0) / max(s.get("SQ_WAVE_CYCLES", 1), 1), 3),
                          "active_frac": round(s.get("SQ_ACTIVE_INST_ANY", 0) / max(s.get("SQ_WAVE_CYCLES", 1), 1), 3)}
  return {"command": "rocprofv3 --kernel-trace --stats -- %s --steps 40 --warmup 10 (5 windows)  (+ separate --pmc passes FETCH_SIZE / WRITE_SIZE / "
                     "TCC_HIT_sum TCC_MISS_sum / SQ_*, --steps 10 --warmup 5)" % COMMANDS[w],
          "kernels": kernels}


def main():
  tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
  src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
  # on the GPU box only gpurun_out/ travels back (<= 64 MiB): summarise there into gpurun_out/<dir>, copy into profiles/ at home
  out_dir = os.path.join(ROOT, sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
  os.makedirs(out_dir, exist_ok=True)
  summary = {"round": int(re.sub(r"\D", "", tag) or 0), "script": "scripts/profile_round.sh + scripts/summarize_profile.py", "workloads": {},
             "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads). "
                     "Kernel averages mix the timed steps with the per-kernel timing loops of bench.py (same kernels, same shapes) and, for "
                     "insert_unique_kernel / insert_evict_kernel, are the 4 M-key pre-fill launches.  sq_per_wave: SQ_* counters per wave "
                     "(wave_quad_cycles in 4-cycle units; wait_any = parked on s_waitcnt, issue_stall = dependency / pipe stalls)."}
  for w in WORKLOADS:
    r = workload(src, tag, w, out_dir)
    if r:
      summary["workloads"][w] = r
  with open(os.path.join(out_dir, tag + "_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
  for w, r in summary["workloads"].items():
    print(w, json.dumps({k: (v["avg_us"], v.get("hbm_bytes_per_launch_corrected")) for k, v in r["kernels"].items()}))


if __name__ == "__main__":
  main()
