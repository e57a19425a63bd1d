"""Do kernels of two HIP streams overlap on this GPU?  find (lookup) on one stream, plan build on another."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import bench
import tfra_amd.dynamic_embedding as de

B, DIM, N = 131072, 64, 10_000_000
opt = de.optimizers.Adam(1e-3)
deo = de.DynamicEmbeddingOptimizer(opt)
var = de.Variable(dim=DIM, name="ovl", initializer=0.0, init_size=int(N * 1.05), **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
for lo in range(0, N, 2_000_000):
  k = bench.keys_of_ranks_torch(torch, torch.arange(lo + 1, lo + 2_000_001, device="cuda"))
  var.upsert(k, torch.zeros((k.numel(), DIM), device="cuda"))
rng = np.random.default_rng(0)
ids = [torch.from_numpy(bench.keys_of_ranks(bench.zipf_bounded(rng, B, N))).cuda() for _ in range(4)]
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
plan = deo.plan(var, ids[0])
torch.cuda.synchronize()
R = 40


def run(do_find, do_plan):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for i in range(R):
    if do_plan:
      with torch.cuda.stream(side):
        plan.build(ids[i & 3], sync=False)
    if do_find:
      var.lookup(ids[(i + 1) & 3])
  main.wait_stream(side)
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) * 1e3 / R, (time.perf_counter() - t0) * 1e6 / R


for name, f, p in (("find only", 1, 0), ("plan only", 0, 1), ("both (2 streams)", 1, 1)):
  run(f, p)
  g, w = run(f, p)
  print("%-18s gpu %7.1f us/iter   wall %7.1f us/iter" % (name, g, w))

# ---- gradient half (tile_sums -> bucket_sums -> apply) on the main stream vs plan build on the side stream
g = torch.randn((B, DIM), device="cuda") * 0.01
planX = deo.plan(var, ids[0])
planY = deo.plan(var, ids[1])
p = opt.params(1)
tab = var.tables[0]._table
dflt = var.tables[0]._default_value.to(torch.float32)
torch.cuda.synchronize()


def run2(do_apply, do_plan):
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  a.record()
  for i in range(R):
    if do_plan:
      with torch.cuda.stream(side):
        planY.build(ids[1 + (i & 1)], sync=False)
    if do_apply:
      tab.apply_planned(p, planX, g, dflt, sync=False)
  main.wait_stream(side)
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) * 1e3 / R, (time.perf_counter() - t0) * 1e6 / R


for name, f, q in (("apply_planned only", 1, 0), ("plan only", 0, 1), ("both (2 streams)", 1, 1)):
  run2(f, q)
  gq, w = run2(f, q)
  print("%-18s gpu %7.1f us/iter   wall %7.1f us/iter" % (name, gq, w))
