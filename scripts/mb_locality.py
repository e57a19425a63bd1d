"""Does the ORDER of the keys matter to find / insert_or_assign on a 10^9-slot table?  Same keys, (a) random order,
(b) sorted by their home bucket (high word of fmix64(key), what tfra_device.h::bucket0 reduces).
  python scripts/mb_locality.py [slots]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  sys.path.insert(0, p)
import numpy as np, torch
import tfra_amd.dynamic_embedding as de
from bench import keys_of_ranks_torch, raw_calls, Timer, SEED

dev = torch.device("cuda", 0)
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dim, dtype, B = 128, torch.float16, 131072
t = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots, max_capacity=slots, device="cuda:0", dim=dim,
                    evict_strategy=de.HkvEvictStrategy.LRU, name="loc")
vals = (torch.randn((4_000_000, dim), device=dev) * 0.01).to(dtype)
for lo in range(1, slots + 1, 4_000_000):
  k = keys_of_ranks_torch(torch, torch.arange(lo, min(slots, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
  t._table.upsert(k, vals[:k.numel()], unique_keys=True)
gen = torch.Generator(device=dev).manual_seed(3)
tm = Timer(torch)
rc = raw_calls(torch, dev)
tbl = t._table
out = torch.empty((B, dim), dtype=dtype, device=dev)


def home_hi(keys):   # high 32 bits of fmix64(key) as a sortable int64
  h = keys_of_ranks_torch(torch, keys ^ SEED)     # keys_of_ranks_torch(x) = fmix64(x ^ SEED)
  return (h >> 32) & 0xFFFFFFFF


for name, sort in (("random", False), ("by home bucket", True)):
  batches = []
  for j in range(12):
    r = torch.randint(1, slots + 1, (B,), generator=gen, device=dev)
    k = keys_of_ranks_torch(torch, r)
    if sort:
      k = k[torch.argsort(home_hi(k))]
    batches.append(k.contiguous())
  finds = [rc.find(tbl._h, b, out, tbl._default_value) for b in batches]
  f_us = tm.us(lambda i: finds[i % 12](), reps=24, warm=3)
  import ctypes
  from tfra_amd import _capi
  lib = _capi.lib()
  st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
  ups = [(lambda a=(tbl._h, b.numel(), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(vals.data_ptr()), None, 1, st): _capi.check(lib.tfra_table_insert_or_assign(*a))) for b in batches]
  u_us = tm.us(lambda i: ups[i % 12](), reps=24, warm=3)
  print("%-15s find(B=131072 resident uniform ids) %.1f us | insert_or_assign(unique, resident) %.1f us" % (name, f_us, u_us), flush=True)
