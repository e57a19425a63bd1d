"""Timeline of the look-ahead step on the GPU (both streams): run scripts/host_profile.py under rocprofv3 --kernel-trace and print,
for a few steady-state steps, every kernel with its start offset, duration and queue.
  python scripts/step_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
  rows = list(csv.DictReader(open(f)))
  rows.sort(key=lambda r: int(r["Start_Timestamp"]))
  names = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
  finds = [i for i, r in enumerate(rows) if "find_kernel" in r["Kernel_Name"] or "find_after_rest" in r["Kernel_Name"]]
  if len(finds) < 60:
    continue
  lo, hi = finds[-40], finds[-34]
  t0 = int(rows[lo]["Start_Timestamp"])
  for r in rows[lo:hi + 1]:
    st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f -> %8.1f us  (%5.1f)  q%-3s %s" % (st / 1e3, en / 1e3, (en - st) / 1e3, r.get("Queue_Id", "?"), names(r)))
  per = (int(rows[finds[-10]]["Start_Timestamp"]) - int(rows[finds[-60]]["Start_Timestamp"])) / 50 / 1e3
  print("average step (find to find, 50 steps): %.1f us" % per)
  # duration histogram per kernel over the steady state (a lookup that completes a pending remainder first is bimodal)
  import collections
  by = collections.defaultdict(list)
  for r in rows[finds[-200]:finds[-1]]:
    by[names(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
  for k, v in by.items():
    v.sort()
    print("%-46s n %4d  p10 %6.1f  p50 %6.1f  p90 %6.1f  max %6.1f us" % (k, len(v), v[len(v) // 10], v[len(v) // 2], v[len(v) * 9 // 10], v[-1]))
