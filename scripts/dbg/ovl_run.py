import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import numpy as np, torch
import tfra_amd.dynamic_embedding as de
from tests.test_gpu_overlap import make_dense_table
dim, cap, n = 64, 200_000, 8192
rng = np.random.default_rng(9)
universe = np.arange(1, int(cap * 0.62) + 1, dtype=np.int64) * 6151 + 1
tabs = [make_dense_table(torch, de, cap, dim, universe, "dbg%d" % i) for i in range(2)]
m = 8
ids = [torch.from_numpy(universe[(rng.zipf(1.2, size=n) * 13) % universe.size]).cuda() for _ in range(m + 1)]
vals = [(torch.randn((n, dim), device="cuda") * 0.01) for _ in range(m)]
outs = [torch.full((n, dim), -7.0, device="cuda") for _ in range(m)]
ref0 = [tabs[i]._table.find(ids[0]) for i in range(2)]
print("tables equal before:", torch.equal(ref0[0], ref0[1]))
d0 = de.OverlapAssignStep(tabs[0])
run = d0.make_run(ids[:m], vals, outs, ids_after=ids[m])
run()
torch.cuda.synchronize()
print("stats", d0.stats())
for k in range(m):
  print(k, "untouched rows:", int((outs[k][:, 0] == -7.0).sum()), "eq ref0:", torch.equal(outs[k], ref0[0]) if k == 0 else "")
d1 = de.OverlapAssignStep(tabs[1]).prime(ids[0])
for k in range(m):
  o = d1.step(vals[k], ids[k + 1])
  torch.cuda.synchronize()
  neq = (o != outs[k]).any(dim=1)
  print(k, "rows differing:", int(neq.sum()), "first", neq.nonzero()[:5].flatten().tolist())
