import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/recommenders-addons_amd")
import numpy as np, torch, ctypes
import tfra_amd.dynamic_embedding as de
from tfra_amd import _capi
dim, cap, B = 16, 60_000, 50_000
t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=cap, max_capacity=cap, device="cuda:0", dim=dim,
                    evict_strategy=de.HkvEvictStrategy.LRU, name="ups_bounded")
rng = np.random.default_rng(3)
fresh = 1
for step in range(6):
    old = (rng.zipf(1.2, size=B // 2) % 20_000).astype(np.int64)
    new = np.arange(fresh, fresh + B // 2, dtype=np.int64) + 1_000_000
    fresh += B // 2
    keys = np.concatenate([old, new]); rng.shuffle(keys)
    vals = np.tile((np.arange(B, dtype=np.float32) + step * B)[:, None], (1, dim))
    t._table.upsert_sparse(torch.from_numpy(keys).cuda(), torch.from_numpy(vals).cuda())
    torch.cuda.synchronize()
    n = int(t.size().item())
    print(step, t._table.slot_census())
    try:
        k, v = t.export()
        print(step, "size", n, "exported", k.numel(), "uniq", np.unique(keys).size)
    except Exception as e:
        print(step, "size", n, "ERR", str(e)[:100], "uniq", np.unique(keys).size)
# where did the keys of the last step go?
kk = torch.from_numpy(np.unique(keys)).cuda()
vals_, ex = t._table.find(kk, return_exists=True) if hasattr(t._table, "find") else (None, None)
print("last step: unique", kk.numel(), "found", int(ex.sum().item()))
