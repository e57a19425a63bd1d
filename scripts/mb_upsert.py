"""Times tfra_table_upsert_planned (+ find) on bounded LRU tables of different sizes at the c3 batch shape: tells the
instruction-bound part (small, cache-resident table) from the memory-bound part (10^9 slots).
  python scripts/mb_upsert.py [slots ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  sys.path.insert(0, p)
import numpy as np, torch
import tfra_amd.dynamic_embedding as de
from bench import keys_of_ranks_torch, keys_of_ranks, zipf_bounded, mixed_batches, raw_calls, Timer, SEED

dev = torch.device("cuda", 0)
B, dim, dtype = 131072, 128, torch.float16
for slots in [int(x) for x in (sys.argv[1:] or ["1000000", "100000000", "1000000000"])]:
  t = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots, max_capacity=slots, device="cuda:0",
                      dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="mb%d" % slots)
  vals = (torch.randn((4_000_000, dim), device=dev) * 0.01).to(dtype)
  for lo in range(1, slots + 1, 4_000_000):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(slots, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
    t._table.upsert(k, vals[:k.numel()], unique_keys=True)
  rng = np.random.default_rng(1)
  nb = 40
  ranks, _ = mixed_batches(rng, nb, B, slots, 0.5, slots + 1)
  ids = torch.from_numpy(keys_of_ranks(ranks.reshape(-1)).reshape(nb, B)).to(dev)
  tm = Timer(torch)
  rc = raw_calls(torch, dev)
  tbl = t._table
  out = torch.empty((B, dim), dtype=dtype, device=dev)
  finds = [rc.find(tbl._h, ids[j], out, tbl._default_value) for j in range(nb)]
  find_us = tm.us(lambda i: finds[i % nb](), reps=20, warm=3)
  plans = [de.table_ops.SparsePlan(dev, 0) for _ in range(16)]
  for j, pl in enumerate(plans):
    pl.build(ids[20 + j], sync=False)
  torch.cuda.synchronize()
  U = sum(plans[0].read()[0][k] for k in ("many", "few"))
  ups = [rc.upsert_planned(tbl._h, plans[j], vals[:B]) for j in range(16)]
  ups_us = tm.us(lambda i: ups[i](), reps=12, warm=4)
  print("slots %d: size %d census %s | find %.1f us | upsert_planned(U=%d) %.1f us" % (slots, int(t.size().item()), tbl.slot_census(), find_us, U, ups_us), flush=True)
  del t, tbl, vals, ups, finds, plans
  torch.cuda.empty_cache()
