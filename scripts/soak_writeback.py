"""Soak run of the write-back as kernels of its own (upsert_own_kernel + upsert_rest_kernel over a SET plan's keys, csrc/tfra_csr.hip) under
its two drivers — the look-ahead driver (PrefetchAssignStep: tfra_table_step_prefetch_assign) and plain calls (find + upsert_sparse) —
on a bounded LRU table at capacity, with a stream whose character CHANGES every few dozen steps: batches of resident ids only (the pass
takes the form without claims, picked from the previous write-back's sample), batches that are half never-seen ids (evictions; the form
with the claims up front), sizes up and down, the two sentinel keys.  Every step: the lookup against a plain find taken before the
write-back, and after the write-back every key of the batch must hold the row of its last occurrence; every `--dict-every` steps the
whole table against a dictionary of last writes kept on the device.   python scripts/soak_writeback.py [--steps 2000] [--slots 20000000]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=2000)
  ap.add_argument("--slots", type=int, default=20_000_000)
  ap.add_argument("--dict-every", type=int, default=250)
  ap.add_argument("--seed", type=int, default=1)
  args = ap.parse_args()
  import torch
  import tfra_amd.dynamic_embedding as de
  from bench import keys_of_ranks_torch
  dev = torch.device("cuda", 0)
  dim, cap = 64, args.slots
  t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=cap, max_capacity=cap, device=str(dev), dim=dim,
                      evict_strategy=de.HkvEvictStrategy.LRU, name="soak_wb")
  tbl = t._table
  for lo in range(1, cap + 1, 4_000_000):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(cap, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
    tbl.upsert(k, (k % 997).to(torch.float32)[:, None].repeat(1, dim), unique_keys=True)
  torch.cuda.synchronize()
  gen = torch.Generator(device=dev).manual_seed(args.seed)
  imin = -(2 ** 63)
  fresh = cap + 1
  sizes = [131072, 131072, 65536, 131072, 40000, 131072, 1000, 131072]
  recent = []   # ranks of recently written never-seen ids: drawn again later (hits on young keys)

  def batch(i):
    nonlocal fresh
    n = sizes[(i // 37) % len(sizes)]
    phase = (i // 23) % 3          # 0: resident ids only; 1: half never-seen; 2: a sixteenth never-seen + young keys again
    hot = (torch.rand(n, generator=gen, device=dev) ** 6 * cap).to(torch.int64) + 1
    if phase:
      m = n // 2 if phase == 1 else n // 16
      if m:
        hot[torch.randint(0, n, (m,), generator=gen, device=dev)] = torch.arange(fresh, fresh + m, dtype=torch.int64, device=dev)
        recent.append((fresh, m))
        fresh += m
      if phase == 2 and recent:
        lo, cnt = recent[int(torch.randint(0, len(recent), (1,), generator=gen, device=dev).item())]
        q = min(cnt, n // 8)
        if q:
          hot[torch.randint(0, n, (q,), generator=gen, device=dev)] = lo + torch.randint(0, cnt, (q,), generator=gen, device=dev)
      del recent[:-8]
    k = keys_of_ranks_torch(torch, hot)
    if n > 8:
      k[torch.randint(0, n, (2,), generator=gen, device=dev)] = imin
      k[torch.randint(0, n, (2,), generator=gen, device=dev)] = imin + 1
    return k

  cur, nxt = batch(0), batch(1)
  ps = de.PrefetchAssignStep(t).prime(cur)
  seen_k = torch.empty(0, dtype=torch.int64, device=dev)
  seen_v = torch.empty(0, dtype=torch.float32, device=dev)
  t0 = time.perf_counter()
  bad = 0
  zero = torch.zeros(dim, device=dev)
  for s in range(args.steps):
    ids = cur
    n = ids.numel()
    vals = (torch.arange(n, device=dev, dtype=torch.float32) + 1e6 * ((s % 1000) + 1))[:, None].repeat(1, dim).contiguous()
    ref, rex = tbl.find(ids, zero, return_exists=True)           # the state the step's lookup has to reflect
    plain = (s // 41) % 2 == 1                                   # the driver changes too
    if plain:
      out, ex = tbl.find(ids, zero, return_exists=True)
      tbl.upsert_sparse(ids, vals)
      ps = None
    else:
      if ps is None:
        ps = de.PrefetchAssignStep(t).prime(ids)
      out = ps.step(vals, nxt)
      ex = rex
    if not (torch.equal(ex, rex) and torch.equal(out.reshape(n, dim), ref)):
      bad += 1
      print("step %d: lookup differs from the table's state" % s, flush=True)
    # every key of the batch holds the row of its last occurrence right after the write-back
    uk, inv = torch.unique(ids, return_inverse=True)
    lp = torch.zeros(uk.numel(), dtype=torch.long, device=dev)
    lp.scatter_reduce_(0, inv, torch.arange(n, device=dev), reduce="amax", include_self=False)
    got, gex = tbl.find(uk, zero, return_exists=True)
    if not (bool(gex.all()) and torch.equal(got, vals[lp])):
      bad += 1
      print("step %d (%s): %d keys of the batch missing, %d with another row" % (s, "plain" if plain else "look-ahead", int((~gex).sum()),
                                                                             int((got != vals[lp]).any(dim=1).sum())), flush=True)
    if bad > 3:
      sys.exit(1)
    allk, allv = torch.cat([seen_k, uk]), torch.cat([seen_v, vals[lp, 0]])
    seen_k, inv2 = torch.unique(allk, return_inverse=True)
    last = torch.zeros(seen_k.numel(), dtype=torch.long, device=dev)
    last.scatter_reduce_(0, inv2, torch.arange(allk.numel(), device=dev), reduce="amax", include_self=False)
    seen_v = allv[last]
    if (s + 1) % args.dict_every == 0 or s + 1 == args.steps:
      got, gex = tbl.find(seen_k, zero, return_exists=True)
      okrows = bool(torch.equal(got[gex][:, 0], seen_v[gex])) and bool((got[gex] == got[gex][:, :1]).all())
      c = tbl.slot_census()
      tbl.check_errors()
      print("step %d: %d keys written so far, %.1f %% still resident, rows of the resident ones %s; size %d <= %d, locked %d"
            % (s + 1, seen_k.numel(), 100.0 * float(gex.float().mean()), "OK" if okrows else "WRONG", int(t.size().item()), tbl.capacity(), c["locked"]),
            flush=True)
      if not okrows or c["locked"]:
        sys.exit(1)
      if seen_k.numel() > 30_000_000:   # (keep the dictionary bounded: forget what was evicted)
        seen_k, seen_v = seen_k[gex], seen_v[gex]
    cur, nxt = nxt, batch(s + 2)
  print("soak OK: %d steps in %.1f s, %d mismatching steps" % (args.steps, time.perf_counter() - t0, bad))
  sys.exit(1 if bad else 0)


if __name__ == "__main__":
  main()
