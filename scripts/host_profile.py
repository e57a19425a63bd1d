"""Host-side cost of the look-ahead step (one C call per step): run under `rocprofv3 --hip-trace --stats` to see where the
calling thread's time goes (hipLaunchKernel and friends), next to the GPU time of the same steps.
  python scripts/host_profile.py [slots] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  sys.path.insert(0, p)
import torch
import tfra_amd.dynamic_embedding as de
from bench import keys_of_ranks_torch, IdFactory

dev = torch.device("cuda", 0)
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
B, dim = 131072, 64
t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=slots, max_capacity=slots, device="cuda:0", dim=dim,
                    evict_strategy=de.HkvEvictStrategy.LRU, name="hostprof")
vals = torch.randn((4_000_000, dim), device=dev) * 0.01
for lo in range(1, slots + 1, 4_000_000):
  k = keys_of_ranks_torch(torch, torch.arange(lo, min(slots, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
  t._table.upsert(k, vals[:k.numel()], unique_keys=True)
idf = IdFactory(torch, dev, B, slots, 0.0, slots + 1, 5)
ids = idf.keys(K + 21)
v1 = vals[:B]
ps = de.PrefetchAssignStep(t).prime(ids[0])
for i in range(20):
  ps.step(v1, ids[i + 1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
  ps.step(v1, ids[20 + i + 1])
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("steps %d: host %.1f us/step, total %.1f us/step" % (K, host / K * 1e6, tot / K * 1e6))
