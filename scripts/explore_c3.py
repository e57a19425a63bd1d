"""Exploration of BASELINE configs[2] at full size on one MI355X: does a 10^9-slot bounded table allocate,
how long does the pre-fill take, what do find / upsert(+eviction) / export cost at that size.
Writes a JSON report to gpurun_out/explore_c3.json.   python scripts/explore_c3.py [slots]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

import numpy as np
import torch

import tfra_amd.dynamic_embedding as de
from bench import keys_of_ranks_torch, zipf_bounded, keys_of_ranks

dev = torch.device("cuda", 0)
rep = {}
free, total = torch.cuda.mem_get_info()
rep["hbm_total_bytes"], rep["hbm_free_bytes_at_start"] = total, free
print("HBM total %.1f GB free %.1f GB" % (total / 1e9, free / 1e9), flush=True)
want = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dim, B = 128, 131072
dtype = torch.float16
t = None
for slots in (want, int(want * 0.9), int(want * 0.8), int(want * 0.6)):
  try:
    t0 = time.perf_counter()
    t = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots, max_capacity=slots,
                        device="cuda:0", dim=dim, evict_strategy=de.HkvEvictStrategy.LRU)
    torch.cuda.synchronize()
    rep["slots"] = slots
    rep["create_s"] = time.perf_counter() - t0
    break
  except Exception as e:  # OOM
    rep.setdefault("alloc_failures", []).append({"slots": slots, "error": str(e)[:300]})
    print("alloc failed at", slots, e, flush=True)
assert t is not None
rep["capacity"] = t._table.capacity()
rep["hbm_free_bytes_after_create"] = torch.cuda.mem_get_info()[0]
print("created", rep, flush=True)

gen = torch.Generator(device=dev).manual_seed(1)
vals = (torch.randn((4_000_000, dim), generator=gen, device=dev) * 0.01).to(dtype)
N = int(sys.argv[2]) if len(sys.argv) > 2 else rep["slots"]
chunk = 4_000_000
t0 = time.perf_counter()
sizes = []
for lo in range(1, N + 1, chunk):
  r = torch.arange(lo, min(N, lo + chunk - 1) + 1, dtype=torch.int64, device=dev)
  k = keys_of_ranks_torch(torch, r)
  t._table.upsert(k, vals[:k.numel()], unique_keys=True)
  if ((lo - 1) // chunk) % 25 == 24:
    torch.cuda.synchronize()
    sizes.append((lo + chunk - 1, int(t.size().item()), round(time.perf_counter() - t0, 2)))
    print("prefill", sizes[-1], flush=True)
torch.cuda.synchronize()
rep["prefill_s"] = time.perf_counter() - t0
rep["prefill_keys"] = N
rep["size_after_prefill"] = int(t.size().item())
rep["prefill_progress"] = sizes
print("prefill done", rep["prefill_s"], rep["size_after_prefill"], flush=True)

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=20, warm=3):
  for i in range(warm):
    fn(i)
  e0.record()
  for i in range(reps):
    fn(warm + i)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


rng = np.random.default_rng(7)
nb = 24
zipf = torch.from_numpy(keys_of_ranks(zipf_bounded(rng, nb * B, N)).reshape(nb, B)).to(dev)
rep["find_zipf_resident_us"] = timed(lambda i: t.lookup(zipf[i % nb]))
fresh0 = N + 1
mixed = []
for i in range(nb):
  new = keys_of_ranks_torch(torch, torch.arange(fresh0 + i * (B // 2), fresh0 + (i + 1) * (B // 2), dtype=torch.int64, device=dev))
  mixed.append(torch.cat([zipf[i, :B // 2], new]))
rep["find_half_unseen_us"] = timed(lambda i: t.lookup(mixed[i % nb]))
uni = torch.from_numpy(keys_of_ranks(rng.integers(1, N + 1, size=nb * B)).reshape(nb, B)).to(dev)
rep["find_uniform_resident_us"] = timed(lambda i: t.lookup(uni[i % nb]))
# write-back halves: unique resident keys (assign) and never-seen keys (insert + eviction)
uq = [torch.unique(zipf[i]) for i in range(nb)]
rep["unique_keys_per_zipf_batch"] = int(np.mean([u.numel() for u in uq]))
v1 = vals[:B]
rep["upsert_zipf_unique_resident_us"] = timed(lambda i: t._table.upsert(uq[i % nb], v1[:uq[i % nb].numel()], unique_keys=True))
news = [keys_of_ranks_torch(torch, torch.arange(fresh0 + i * B, fresh0 + (i + 1) * B, dtype=torch.int64, device=dev)) for i in range(64)]
sz0 = int(t.size().item())
rep["upsert_all_new_131072_us"] = timed(lambda i: t._table.upsert(news[i], v1, unique_keys=True), reps=40, warm=4)
rep["size_before_after_new_upserts"] = [sz0, int(t.size().item())]
# the c3 step as two plain calls: find(B mixed) + upsert(unique(mixed))
fresh1 = fresh0 + 64 * B + 1   # ranks nobody has touched yet
mixed = []
for i in range(nb):
  new = keys_of_ranks_torch(torch, torch.arange(fresh1 + i * (B // 2), fresh1 + (i + 1) * (B // 2), dtype=torch.int64, device=dev))
  mixed.append(torch.cat([zipf[i, :B // 2], new]))
mu = [torch.unique(m) for m in mixed]
rep["c3_unique_keys_per_batch"] = int(np.mean([u.numel() for u in mu]))


def c3_step(i):
  m = mixed[i % nb]
  t.lookup(m)
  u = mu[i % nb]
  t._table.upsert(u, v1[:u.numel()], unique_keys=True)


rep["c3_step_find_plus_unique_upsert_us"] = timed(c3_step, reps=nb - 3, warm=3)
rep["tfra_unique_us"] = timed(lambda i: de.device_ops.unique(mixed[i % nb]))
# export sweep: one window of 16M slots
from tfra_amd import _capi
from tfra_amd.dynamic_embedding.table_ops import _ptr, _stream
win = 16_000_000
kbuf = torch.empty(win, dtype=torch.int64, device=dev)
vbuf = torch.empty((win, dim), dtype=dtype, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def exp(i):
  cnt.zero_()
  _capi.call("tfra_table_export_batch", t._table._h, win, (i % 8) * win, _ptr(cnt), _ptr(kbuf), _ptr(vbuf), None,
             _stream(t._table.device))


rep["export_16M_slot_window_us"] = timed(exp, reps=8, warm=1)
rep["export_window_live_keys"] = int(cnt.item())
rep["size_at_end"] = int(t.size().item())
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "explore_c3.json"), "w"), indent=1)
print(json.dumps(rep, indent=1))
