"""Times TFRA>HkvHashTableEmbeddingLookup's calls on a bounded table at capacity (dim 64 fp32, B = 131072 Zipf-1.2): tfra_table_find, then
tfra_unique_unordered, against tfra_table_find_unique (both in one launch), and each followed by tfra_table_insert_or_assign_n.
  python scripts/mb_find_unique.py [--slots 200000000]"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--slots", type=int, default=200_000_000)
  args = ap.parse_args()
  import torch
  import tfra_amd.dynamic_embedding as de
  from tfra_amd import _capi
  from tfra_amd.dynamic_embedding.device_ops import _workspace
  from tfra_amd.dynamic_embedding.table_ops import _stream
  from bench import IdFactory, keys_of_ranks_torch, SEED
  dev = torch.device("cuda", 0)
  B, dim, cap = 131072, 64, args.slots
  t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=cap, max_capacity=cap, device=str(dev), dim=dim,
                      evict_strategy=de.HkvEvictStrategy.LRU, name="mb_fu")
  tbl = t._table
  vals_fill = torch.randn((4_000_000, dim), device=dev) * 0.01
  for lo in range(((cap - 1) // 4_000_000) * 4_000_000 + 1, 0, -4_000_000):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(cap, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
    tbl.upsert(k, vals_fill[:k.numel()], unique_keys=True)
  del vals_fill
  idf = IdFactory(torch, dev, B, cap, 0.0, cap + 1, SEED + 7)
  NB = 64
  ids = idf.keys(NB)
  lib = _capi.lib()
  P = lambda x: ctypes.c_void_p(x.data_ptr())
  st = _stream(dev)
  ws = _workspace(dev)
  out = torch.empty((B, dim), device=dev)
  values = torch.randn((B, dim), device=dev)
  ub = torch.empty(B, dtype=torch.int64, device=dev)
  ib = torch.empty(B, dtype=torch.int32, device=dev)
  cnt = torch.zeros((), dtype=torch.int64, device=dev)
  dflt = tbl._default_value

  def find(i):
    _capi.check(lib.tfra_table_find(tbl._h, B, P(ids[i]), P(out), None, P(dflt), 0, st))

  def uniq(i):
    _capi.check(lib.tfra_unique_unordered(ws, B, P(ids[i]), P(ub), P(ib), P(cnt), st))

  def fu(i):
    _capi.check(lib.tfra_table_find_unique(tbl._h, ws, B, P(ids[i]), P(out), None, P(dflt), 0, P(ub), P(ib), P(cnt), st))

  def ins(i):
    _capi.check(lib.tfra_table_insert_or_assign_n(tbl._h, B, P(cnt), P(ub), P(values), None, st))

  def timed(name, fn):
    for i in range(8):
      fn(i)
    ts = []
    for w in range(5):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for i in range(NB):
        fn(i)
      torch.cuda.synchronize()
      ts.append((time.perf_counter() - t0) / NB * 1e6)
    print("%-46s %6.2f us per batch (windows %s)" % (name, sorted(ts)[2], " ".join("%.1f" % x for x in ts)), flush=True)

  timed("tfra_table_find", find)
  timed("tfra_unique_unordered", uniq)
  timed("find, then unique_unordered", lambda i: (find(i), uniq(i)))
  timed("tfra_table_find_unique (one launch)", fu)
  timed("find, unique_unordered, insert_or_assign_n", lambda i: (find(i), uniq(i), ins(i)))
  timed("find_unique, insert_or_assign_n", lambda i: (fu(i), ins(i)))
  # same results
  fu(3)
  a, u1 = out.clone(), int(cnt.item())
  s1 = torch.sort(ub[:u1]).values.clone()
  ok_idx = bool(torch.equal(ub[ib.long()], ids[3]))
  find(3); uniq(3)
  print("same rows:", bool(torch.equal(a, out)), " same distinct ids:", u1 == int(cnt.item()) and bool(torch.equal(s1, torch.sort(ub[:u1]).values)),
        " inverse index consistent:", ok_idx)
  tbl.check_errors()


if __name__ == "__main__":
  main()
