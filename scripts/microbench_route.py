"""Per-op cost of the multi-GPU id route on ONE GPU (B = 131072 Zipf-1.2 ids, dim 64): what each
device op of AllToAllEmbedding costs, so the N>1 step can be budgeted without an 8-GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import bench  # noqa: E402
import tfra_amd.dynamic_embedding as de  # noqa: E402

B, DIM, WORLD = 131072, 64, 8
rng = np.random.default_rng(1)
ids = torch.from_numpy(bench.keys_of_ranks(bench.zipf_bounded(rng, B, 10**8))).cuda()
g = torch.randn((B, DIM), device="cuda") * 0.01
ops = de.device_ops


def timeit(name, fn, reps=50):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  print("%-34s gpu %8.1f us   wall %8.1f us" % (name, a.elapsed_time(b) * 1e3 / reps, (time.perf_counter() - t0) * 1e6 / reps))


uniq, idx, cnt = ops.unique_no_sync(ids)
u = int(cnt.item())
print("unique", u, "of", B)
om, perm, counts = ops.partition(uniq, WORLD, 0, n_dev=cnt)
rows_u = torch.randn((u, DIM), device="cuda")
timeit("unique_no_sync", lambda: ops.unique_no_sync(ids))
timeit("partition(uniq, n_dev)", lambda: ops.partition(uniq, WORLD, 0, n_dev=cnt))
timeit("partition(all ids)", lambda: ops.partition(ids, WORLD, 0))
timeit("counts .tolist() (host sync)", lambda: torch.stack([counts, counts]).tolist())
timeit("scatter_rows u", lambda: ops.scatter_rows(rows_u, perm[:u]))
timeit("gather_rows B <- u", lambda: ops.gather_rows(rows_u, idx))
timeit("gather_rows B <- B", lambda: ops.gather_rows(g, perm))
timeit("reduce_by_key", lambda: ops.reduce_by_key(ids, g))
timeit("torch.empty x4", lambda: [torch.empty((B, DIM), device="cuda") for _ in range(4)])
