"""Host cost of one look-ahead step (tfra_table_step_prefetch_assign through PrefetchAssignStep.step) with kernels that take
no time: a 64 K-slot table and 256-id batches — what is left is the calling thread's work per step (four launches, one event
record and query, the Python wrapper).  If this is close to the step time of the real workload, the host bounds it.
  python scripts/host_only.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  sys.path.insert(0, p)
import torch
import tfra_amd.dynamic_embedding as de
dev = torch.device("cuda", 0)
dim, B, K = 64, 256, 2000
t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=1 << 16, max_capacity=1 << 16, device="cuda:0", dim=dim,
                    evict_strategy=de.HkvEvictStrategy.LRU, name="host_only")
ids = [torch.randint(0, 1 << 14, (B,), device=dev, dtype=torch.int64) for _ in range(64)]
vals = torch.randn((B, dim), device=dev)
ps = de.PrefetchAssignStep(t).prime(ids[0])
for i in range(200):
  ps.step(vals, ids[(i + 1) % 64])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
  ps.step(vals, ids[(i + 1) % 64])
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("tiny batches: host %.1f us/step, total %.1f us/step" % (host / K * 1e6, tot / K * 1e6))
t0 = time.perf_counter()
for i in range(K):
  ps.step(vals, ids[(i + 1) % 64], lookup=False)
host = time.perf_counter() - t0
torch.cuda.synchronize()
print("  without the lookup (no output tensor, three launches): host %.1f us/step" % (host / K * 1e6))
