"""What the owner's launch of the routed step costs ALONE: tfra_table_step_overlap on the DISTINCT ids of Zipf-1.2 batches (what a rank
serves at one rank) against the same on the full batches — same table, one stream, nothing else running.
  python scripts/mb_owner_step.py [--slots N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--slots", type=int, default=500_000_000)
  ap.add_argument("--steps", type=int, default=64)
  args = ap.parse_args()
  import torch
  import tfra_amd.dynamic_embedding as de
  from bench import IdFactory, keys_of_ranks_torch, SEED
  dev = torch.device("cuda", 0)
  B, dim, dtype = 131072, 64, torch.float32
  slots = args.slots
  table = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots, max_capacity=slots, device=str(dev), dim=dim,
                          evict_strategy=de.HkvEvictStrategy.LRU, name="mb_owner")
  gen = torch.Generator(device=dev).manual_seed(SEED)
  chunk = 4_000_000
  vals_fill = (torch.randn((chunk, dim), generator=gen, device=dev) * 0.01).to(dtype)
  for lo in range(((slots - 1) // chunk) * chunk + 1, 0, -chunk):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(slots, lo + chunk - 1) + 1, dtype=torch.int64, device=dev))
    table._table.upsert(k, vals_fill[:k.numel()], unique_keys=True)
  print("resident", int(table.size().item()), flush=True)
  del vals_fill
  K = args.steps
  idf = IdFactory(torch, dev, B, slots, 0.0, slots + 1, SEED + 7)
  full = idf.keys(3 * K + 8)
  values = (torch.randn((B, dim), generator=gen, device=dev) * 0.01).to(dtype)
  for name, ids in (("full batches (131072 ids)", [full[i] for i in range(3 * K + 8)]),
                    ("distinct ids only", [torch.unique(full[i]) for i in range(3 * K + 8)]),
                    ("distinct ids, shuffled", [torch.unique(full[i])[torch.randperm(torch.unique(full[i]).numel(), device=dev)] for i in range(3 * K + 8)])):
    ovl = de.OverlapAssignStep(table)
    ovl.prime(ids[0])
    for i in range(K):
      ovl.step(values[:ids[i].numel()], ids[i + 1], ids[i + 2])
    ts = []
    for w in range(2):
      ovl.time_kernels(K)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for i in range(K * (w + 1), K * (w + 2)):
        ovl.step(values[:ids[i].numel()], ids[i + 1], ids[i + 2])
      torch.cuda.synchronize()
      ts.append((time.perf_counter() - t0) / K * 1e6)
      kt = ovl.kernel_times()
    if os.environ.get("TFRA_STEP_VARIANT") and int(os.environ["TFRA_STEP_VARIANT"]) & 16:
      tm = ovl.timing()
      roles = ("build", "scatter", "write-back", "lookup", "tail", "map")
      for row in tm[-3:]:
        print("   " + "  ".join("%s %s" % (roles[r], "-" if x is None else "%.1f..%.1f med %.1f p95 %.1f" % x) for r, x in enumerate(row)), flush=True)
    ovl.flush()
    st = ovl.stats()
    print("%-28s n ~ %6d: %.1f / %.1f us per step, launch %.1f us (events), overlapped %d sequential %d listed %d" %
          (name, ids[0].numel(), ts[0], ts[1], kt["step_kernel_us"], st["overlapped"], st["sequential"], st["lookups_listed"]), flush=True)
    del ovl


if __name__ == "__main__":
  main()
