"""Host-side cost of one C-ABI call vs device time: python scripts/microbench_host.py"""
import os, sys, ctypes, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import torch
import tfra_amd.dynamic_embedding as de
from tfra_amd import _capi
from tfra_amd.dynamic_embedding.table_ops import _ptr, _stream
from bench import zipf_bounded, keys_of_ranks
from scripts.microbench import timeit, build
dev = torch.device("cuda:0"); B = 131072
lib = _capi.lib()
dbg = lib.tfra_debug_find_variant
dbg.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t] + [ctypes.c_void_p] * 4
n = 10_000_000
t = build(n, 2)
rng = np.random.default_rng(0)
ids = torch.from_numpy(keys_of_ranks(zipf_bounded(rng, B, n))).to(dev)
obuf = torch.empty((B, 64), device=dev); dflt = torch.zeros(64, device=dev); st = _stream(dev)
fn = lib.tfra_table_find
a1 = (t._table._h, B, _ptr(ids), _ptr(obuf), None, _ptr(dflt), 0, st)
a2 = (t._table._h, 0, 4, B, _ptr(ids), _ptr(obuf), _ptr(dflt), st)
for name, f, a in [("tfra_table_find", fn, a1), ("debug variant", dbg, a2)]:
  for B_ in (B, 1024):
    aa = list(a); aa[1 if f is fn else 3] = B_
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): f(*aa)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-16s n=%6d host issue %.2f us/call, incl. drain %.2f us/call" % (name, B_, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
