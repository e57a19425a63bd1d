"""Ablation of the find kernel: python scripts/microbench_find.py"""
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import torch
import tfra_amd.dynamic_embedding as de
from tfra_amd import _capi
from tfra_amd.dynamic_embedding.table_ops import _ptr, _stream
from bench import zipf_bounded, keys_of_ranks, keys_of_ranks_torch
from scripts.microbench import timeit, build

dev = torch.device("cuda:0"); B = 131072
lib = _capi.lib()
dbg = lib.tfra_debug_find_variant
dbg.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t] + [ctypes.c_void_p] * 4
rng = np.random.default_rng(0)
n = int(os.environ.get("NKEYS", 100_000_000))
t = build(n, 2)
zipf = torch.from_numpy(keys_of_ranks(zipf_bounded(rng, B, n))).to(dev)
uni = torch.from_numpy(keys_of_ranks(rng.choice(n, B, replace=False).astype(np.int64) + 1)).to(dev)
obuf = torch.empty((B, 64), device=dev); dflt = torch.zeros(64, device=dev); st = _stream(dev)
src = torch.randn((B, 64), device=dev)
print("torch copy 33.5MB: %.1fus" % timeit(lambda: obuf.copy_(src), reps=100))
for name, ids in [("zipf", zipf), ("uniform", uni)]:
  for mode, mname in [(0, "full"), (1, "gather+store,no probe"), (2, "probe only"), (3, "store zeros"), (4, "first line only"), (7, "prod copy")] + [(10 + i, "fk clamp=%d bar=%d ex=%d" % (i & 1, (i >> 1) & 1, (i >> 2) & 1)) for i in range(8)]:
    r = []
    for U in ((1, 2, 4, 8) if mode < 7 else (4,)):
      args = (t._table._h, mode, U, B, _ptr(ids), _ptr(obuf), _ptr(dflt), st)
      r.append("U%d=%.1f" % (U, timeit(lambda: dbg(*args), reps=200)))
    print("%-8s %-24s %s" % (name, mname, "  ".join(r)), flush=True)
