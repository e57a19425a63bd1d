"""The reference's Insert op (tfra_table_insert_or_assign with TFRA_FLAG_UNIQUE_KEYS: upsert_own_kernel<…, DIRECT> + upsert_rest_kernel) on a
GROWING table (TFRA's default cuckoo flavour), per call by HIP events: the unique keys of Zipf-1.2 batches over the resident keys, with
`new_ratio` of a batch never seen before.   TFRA_OWN_HF=0 python scripts/mb_insert_growing.py [keys] [new_ratio]   (A/B of the pass's form)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  sys.path.insert(0, p)
import numpy as np, torch
import tfra_amd.dynamic_embedding as de
from bench import keys_of_ranks_torch, keys_of_ranks, mixed_batches, Timer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
new_ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda", 0)
B, dim = 131072, 64
t = de.CuckooHashTable(torch.int64, torch.float32, torch.zeros(dim), device="cuda:0", dim=dim, name="mbi", init_size=int(N * 1.3))
vals = torch.randn((4_000_000, dim), device=dev) * 0.01
for lo in range(1, N + 1, 4_000_000):
  k = keys_of_ranks_torch(torch, torch.arange(lo, min(N, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
  t._table.upsert(k, vals[:k.numel()], unique_keys=True)
rng = np.random.default_rng(1)
nb = 64
ranks, _ = mixed_batches(rng, nb, B, N, new_ratio, N + 1)
uq = [torch.unique(torch.from_numpy(keys_of_ranks(ranks[j])).to(dev)) for j in range(nb)]
tbl = t._table
tm = Timer(torch)
import ctypes
from tfra_amd import _capi
lib = _capi.lib()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda x: ctypes.c_void_p(x.data_ptr())
calls = [(tbl._h, uq[j].numel(), P(uq[j]), P(vals), None, 1, st) for j in range(nb)]   # (pre-built arguments: a Python-level call costs more than the kernels)
for rnd in range(3):
  us = tm.us(lambda i: _capi.check(lib.tfra_table_insert_or_assign(*calls[i % nb])), reps=48, warm=8)
  print("TFRA_OWN_HF=%s  keys %d  new_ratio %.2f  U~%d: insert_or_assign(unique keys) %.2f us per call" % (
      os.environ.get("TFRA_OWN_HF", "auto"), N, new_ratio, uq[0].numel(), us), flush=True)
tbl.check_errors()
