"""BASELINE configs[4] shape on ONE GPU (a parity-test configuration, not the bench line): 26 tables, dims cycling
{16,32,64,128}, one id per table per sample, batch 131072, FTRL — step time with plain calls vs one
PrefetchStep per table (each table's next-batch plan built on its own second stream)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import bench
import tfra_amd.dynamic_embedding as de

NT, B, STEPS, WARM = 26, 131072, 30, 5
dims = [16, 32, 64, 128]
sizes = np.unique(np.logspace(5.5, 7.0, NT).astype(np.int64))          # 0.3 M .. 10 M keys (scaled to one GPU)
sizes = np.resize(sizes, NT)
rng = np.random.default_rng(0)
results = {}
for mode in ("plain", "prefetch"):
  tabs = []
  for i in range(NT):
    opt = de.optimizers.Ftrl(0.05, l1_regularization_strength=1e-3, l2_regularization_strength=1e-3)
    deo = de.DynamicEmbeddingOptimizer(opt)
    v = de.Variable(dim=dims[i % 4], name="c5_%s_%d" % (mode, i), initializer=0.0, init_size=int(sizes[i] * 1.1),
                    **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    n = int(sizes[i])
    for lo in range(0, n, 2_000_000):
      k = bench.keys_of_ranks_torch(torch, torch.arange(lo + 1, min(lo + 2_000_000, n) + 1, device="cuda"))
      v.upsert(k, torch.zeros((k.numel(), v.dim), device="cuda"))
    ids = [torch.from_numpy(bench.keys_of_ranks(bench.zipf_bounded(rng, B, n))).cuda() for _ in range(4)]
    g = torch.randn((B, v.dim), device="cuda") * 0.01
    ps = de.PrefetchStep(v, deo).prime(ids[0]) if mode == "prefetch" else None
    tabs.append((v, deo, ids, g, ps))
  torch.cuda.synchronize()

  def step(s):
    for v, deo, ids, g, ps in tabs:
      if ps is not None:
        ps.step(g, ids[(s + 1) & 3])
      else:
        v.lookup(ids[s & 3])
        deo.apply_sparse(v, ids[s & 3], g)

  for s in range(WARM):
    step(s)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for s in range(STEPS):
    step(WARM + s)
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) * 1e3 / STEPS
  results[mode] = ms
  print("%-9s %.3f ms per step of 26 tables  = %.2f G lookup+insert pairs/s" % (mode, ms, NT * B / ms / 1e6))
  del tabs
  torch.cuda.empty_cache()
