"""Micro-benchmarks of single kernels (not the headline bench): python scripts/microbench.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import torch
import tfra_amd.dynamic_embedding as de
from bench import zipf_bounded, keys_of_ranks, keys_of_ranks_torch

dev = torch.device("cuda:0")
B = 131072

def timeit(fn, reps=30):
  for _ in range(5): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps

def build(n, aux):
  t = de.CuckooHashTable(torch.int64, torch.float32, torch.zeros(64), init_size=int(n * 1.05), device="cuda:0", dim=64,
                         aux_fields=aux)
  for lo in range(1, n + 1, 4_000_000):
    r = torch.arange(lo, min(n, lo + 3_999_999) + 1, dtype=torch.int64, device=dev)
    t._table.upsert(keys_of_ranks_torch(torch, r), torch.randn((r.numel(), 64), device=dev) * 0.01, unique_keys=True)
  return t

rng = np.random.default_rng(0)
for n in [1_000_000, 10_000_000, 100_000_000]:
  for aux in [0, 2]:
    t = build(n, aux)
    zipf = torch.from_numpy(keys_of_ranks(zipf_bounded(rng, B, n))).to(dev)
    uni = torch.from_numpy(keys_of_ranks(rng.choice(n, B, replace=False).astype(np.int64) + 1)).to(dev)
    miss = torch.from_numpy(rng.integers(2**62, 2**63 - 1, B)).to(dev)
    out = {}
    from tfra_amd import _capi
    from tfra_amd.dynamic_embedding.table_ops import _ptr, _stream
    obuf = torch.empty((B, 64), device=dev)
    dflt = torch.zeros(64, device=dev)
    st = _stream(dev)
    fn = _capi.lib().tfra_table_find
    for name, ids in [("zipf", zipf), ("uniform-unique", uni), ("all-miss", miss)]:
      out[name + "(py)"] = timeit(lambda: t.lookup(ids))
      args = (t._table._h, B, _ptr(ids), _ptr(obuf), None, _ptr(dflt), 0, st)
      out[name + "(C)"] = timeit(lambda: fn(*args), reps=200)
    vals = torch.randn((B, 64), device=dev)
    out["upsert-unique(resident)"] = timeit(lambda: t._table.upsert(uni, vals, unique_keys=True))
    out["upsert-dupsafe(zipf)"] = timeit(lambda: t._table.upsert(zipf, vals))
    print("keys=%d aux=%d  " % (n, aux) + "  ".join("%s=%.1fus" % kv for kv in out.items()), flush=True)
    del t
    torch.cuda.empty_cache()
