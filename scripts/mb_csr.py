"""Per-kernel timings of the CSR write-back WITHOUT stream overlap (run under rocprofv3 --kernel-trace):
plan builds, gradient halves, lookups, each phase alone.   python scripts/mb_csr.py [keys] [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)
import numpy as np
import torch
import tfra_amd.dynamic_embedding as de
from tfra_amd.dynamic_embedding.table_ops import SparsePlan
from bench import keys_of_ranks_torch, zipf_bounded, keys_of_ranks

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B, dim = (int(sys.argv[3]) if len(sys.argv) > 3 else 131072), 64
dev = torch.device("cuda", 0)
opt = de.optimizers.Adam(1e-3)
deo = de.DynamicEmbeddingOptimizer(opt)
var = de.Variable(dim=dim, name="mb", initializer=0.0, init_size=int(N * 1.05), **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
t = var.tables[0]
gen = torch.Generator(device=dev).manual_seed(1)
for lo in range(1, N + 1, 4_000_000):
  k = keys_of_ranks_torch(torch, torch.arange(lo, min(N, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
  t._table.upsert(k, torch.randn((k.numel(), dim), generator=gen, device=dev) * 0.01, unique_keys=True)
rng = np.random.default_rng(3)
ids = torch.from_numpy(keys_of_ranks(zipf_bounded(rng, reps * B, N)).reshape(reps, B)).to(dev)
grads = torch.randn((B, dim), generator=gen, device=dev) * 0.01
plans = [SparsePlan(dev, dim) for _ in range(reps)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
p = opt.params(1)
dflt = t._default_value.to(torch.float32)


def phase(name, fn):
  torch.cuda.synchronize()
  e0.record()
  for i in range(reps):
    fn(i)
  e1.record()
  torch.cuda.synchronize()
  print("%-28s %8.2f us per call" % (name, e0.elapsed_time(e1) * 1e3 / reps), flush=True)


for rnd in range(2):
  phase("plan build", lambda i: plans[i].build(ids[i], sync=False))
  phase("gradient half (planned)", lambda i: t._table.apply_planned(p, plans[i], grads, dflt, sync=False))
  phase("lookup", lambda i: t.lookup(ids[i]))
  phase("apply_sparse (one call)", lambda i: t._table.apply_sparse(p, ids[i], grads, dflt))
counts = plans[0].read()[0]
print(counts)
