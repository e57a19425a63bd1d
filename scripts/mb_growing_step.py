"""The overlapped step on a GROWING table (CuckooHashTable: TFRA's default creator): lookup + insert_or_assign of Zipf-1.2 batches over 10^8
resident keys, `--new-key-ratio` of every batch never seen.  TFRA_STEP_GROWING=0 = the sequential path inside the same entry point.
  python scripts/mb_growing_step.py [--keys N] [--new-key-ratio r]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--keys", type=int, default=100_000_000)
  ap.add_argument("--new-key-ratio", type=float, default=0.0)
  ap.add_argument("--steps", type=int, default=100)
  args = ap.parse_args()
  import torch
  import tfra_amd.dynamic_embedding as de
  from bench import IdFactory, keys_of_ranks_torch, SEED
  dev = torch.device("cuda", 0)
  B, dim = 131072, 64
  t = de.CuckooHashTable(torch.int64, torch.float32, torch.zeros(dim), device=str(dev), dim=dim, name="mb_grow", init_size=int(args.keys * 1.4))
  vals_fill = torch.randn((4_000_000, dim), device=dev) * 0.01
  for lo in range(1, args.keys + 1, 4_000_000):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(args.keys, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
    t._table.upsert(k, vals_fill[:k.numel()], unique_keys=True)
  del vals_fill
  K = args.steps
  idf = IdFactory(torch, dev, B, args.keys, args.new_key_ratio, args.keys + 1, SEED + 7)
  ids = idf.keys(2 * K + 8)
  values = torch.randn((B, dim), device=dev) * 0.01
  torch.cuda.synchronize()
  ovl = de.OverlapAssignStep(t).prime(ids[0])
  for i in range(8):
    ovl.step(values, ids[i + 1], ids[i + 2])
  runs = [ovl.make_run([ids[8 + i]], [values], [torch.empty((B, dim), device=dev)], ids_after=ids[9 + i], values_before=values, ids_after2=ids[10 + i]) for i in range(2 * K - 4)]
  ts = []
  for w in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(w * (K - 2), (w + 1) * (K - 2)):
      runs[i]()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / (K - 2) * 1e6)
  ovl.flush()
  st = ovl.stats()
  t._table.check_errors()
  print("TFRA_STEP_GROWING=%s new %.2f: %.1f / %.1f us per step, overlapped %d sequential %d, size %d capacity %d" %
        (os.environ.get("TFRA_STEP_GROWING", "1"), args.new_key_ratio, ts[0], ts[1], st["overlapped"], st["sequential"], int(t.size().item()), t._table.capacity()), flush=True)


if __name__ == "__main__":
  main()
