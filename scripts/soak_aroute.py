"""Soak run of the ROUTED assign step (csrc/tfra_aroute.hip, one rank through the route driver: RoutedAssignStep(transport="local")):
thousands of steps on a bounded LRU table at capacity with a feed depth that comes and goes (0 ... 5 batches ahead), batch sizes that change, never-seen ids (evictions) and the two sentinel keys, every step checked against a plain find of
the table (the state the lookup had to reflect) and, every `--dict-every` steps, the whole table against a dictionary of last writes kept
on the device.   python scripts/soak_aroute.py [--steps 3000] [--slots 20000000]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=3000)
  ap.add_argument("--slots", type=int, default=20_000_000)
  ap.add_argument("--dict-every", type=int, default=250)
  ap.add_argument("--seed", type=int, default=1)
  ap.add_argument("--churn", action="store_true", help="every batch: a quarter never-seen ids, the rest drawn by recency of insertion down to "
                  "the keys about to be evicted — the table turns over every few dozen steps")
  args = ap.parse_args()
  import torch
  import tfra_amd.dynamic_embedding as de
  from bench import keys_of_ranks_torch
  dev = torch.device("cuda", 0)
  dim, cap = 64, args.slots
  t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=cap, max_capacity=cap, device=str(dev), dim=dim,
                      evict_strategy=de.HkvEvictStrategy.LRU, name="soak")
  tbl = t._table
  for lo in range(1, cap + 1, 4_000_000):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(cap, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
    tbl.upsert(k, (k % 997).to(torch.float32)[:, None].repeat(1, dim), unique_keys=True)
  torch.cuda.synchronize()
  gen = torch.Generator(device=dev).manual_seed(args.seed)
  imin = -(2 ** 63)
  fresh = cap + 1
  sizes = [131072, 131072, 65536, 131072, 40000, 131072, 1000, 131072]

  def batch(i):
    nonlocal fresh
    n = sizes[(i // 37) % len(sizes)]
    if args.churn:
      hot = (fresh - 1 - (torch.rand(n, generator=gen, device=dev) ** 2 * 1.1 * cap).to(torch.int64)).clamp_(min=1)
    else:
      hot = (torch.rand(n, generator=gen, device=dev) ** 6 * cap).to(torch.int64) + 1
    if args.churn or i % 3 == 0:   # a share of never-seen ranks: evictions
      m = n // 4 if args.churn else n // 16
      hot[torch.randint(0, n, (m,), generator=gen, device=dev)] = torch.arange(fresh, fresh + m, dtype=torch.int64, device=dev)
      fresh += m
    k = keys_of_ranks_torch(torch, hot)
    if n > 8:
      k[torch.randint(0, n, (2,), generator=gen, device=dev)] = imin
      k[torch.randint(0, n, (2,), generator=gen, device=dev)] = imin + 1
    return k

  from tfra_amd.dynamic_embedding.distributed import RoutedAssignStep
  rs = RoutedAssignStep(t, transport="local", max_batch=1 << 18)
  queue = []          # batches fed and not yet looked up
  made = 0

  def feed_upto(depth):
    nonlocal made
    while len(queue) < depth + 1:
      b = batch(made); made += 1
      rs.feed(b)
      queue.append(b)

  seen_k = torch.empty(0, dtype=torch.int64, device=dev)
  seen_v = torch.empty(0, dtype=torch.float32, device=dev)
  t0 = time.perf_counter()
  bad = 0
  prev_vals = None
  for s in range(args.steps):
    depth = (0, 5, 5, 2, 5, 1, 5, 3)[(s // 11) % 8]     # batches fed ahead of the one being looked up
    feed_upto(depth)
    ids = queue.pop(0)
    n = ids.numel()
    out = rs.step(prev_vals)
    torch.cuda.synchronize()
    ref = tbl.find(ids)
    if not torch.equal(out, ref):
      bad += 1
      print("step %d: lookup differs from the table's state (%d rows)" % (s, int((out != ref).any(dim=1).sum())), flush=True)
      if bad > 3:
        sys.exit(1)
    vals = (torch.arange(n, device=dev, dtype=torch.float32) + 1e6 * ((s % 1000) + 1))[:, None].repeat(1, dim).contiguous()
    prev_vals = vals
    uk, inv = torch.unique(ids, return_inverse=True)
    lp = torch.zeros(uk.numel(), dtype=torch.long, device=dev)
    lp.scatter_reduce_(0, inv, torch.arange(n, device=dev), reduce="amax", include_self=False)
    allk, allv = torch.cat([seen_k, uk]), torch.cat([seen_v, vals[lp, 0]])
    seen_k, inv2 = torch.unique(allk, return_inverse=True)
    last = torch.zeros(seen_k.numel(), dtype=torch.long, device=dev)
    last.scatter_reduce_(0, inv2, torch.arange(allk.numel(), device=dev), reduce="amax", include_self=False)
    seen_v = allv[last]
    if (s + 1) % args.dict_every == 0 or s + 1 == args.steps:
      rs.flush(prev_vals)
      prev_vals = None
      torch.cuda.synchronize()
      got, gex = tbl.find(seen_k, return_exists=True)
      okrows = bool(torch.equal(got[gex][:, 0], seen_v[gex])) and bool((got[gex] == got[gex][:, :1]).all())
      st = rs.stats()
      c = tbl.slot_census()
      tbl.check_errors()
      print("step %d: %d keys written so far, %.1f %% still resident, rows of the resident ones %s; size %d <= %d, locked %d; %s"
            % (s + 1, seen_k.numel(), 100.0 * float(gex.float().mean()), "OK" if okrows else "WRONG", int(t.size().item()), tbl.capacity(), c["locked"], st), flush=True)
      if not okrows or c["locked"]:
        sys.exit(1)
  print("soak OK: %d routed steps in %.1f s, %d mismatching steps" % (args.steps, time.perf_counter() - t0, bad))
  rs.close()
  sys.exit(1 if bad else 0)


if __name__ == "__main__":
  main()
