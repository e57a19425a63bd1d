import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import numpy as np, torch
import tfra_amd.dynamic_embedding as de
from tests.test_gpu_overlap import make_dense_table
dim, cap, n = 64, 200_000, 8192
rng = np.random.default_rng(9)
universe = np.arange(1, int(cap * 0.62) + 1, dtype=np.int64) * 6151 + 1
t = make_dense_table(torch, de, cap, dim, universe, "dbg")
ids = [torch.from_numpy(universe[(rng.zipf(1.2, size=n) * 13) % universe.size]).cuda() for _ in range(10)]
vals = torch.randn((n, dim), device="cuda")
d = de.OverlapAssignStep(t).prime(ids[0])
for k in range(9):
  d.step(vals, ids[k + 1])
  torch.cuda.synchronize()
  print(k, d.stats(), flush=True)
