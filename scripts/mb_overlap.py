"""Tuning run for the overlapped step (csrc/tfra_step_impl.h) on the metric's configuration: 10^9-slot bounded LRU table, dim 64
fp32, B = 131072 Zipf-1.2.  Times, per kernel variant (TFRA_STEP_VARIANT): one C call per step, and D steps per host call
(tfra_table_steps_overlap); next to them the round-3 look-ahead driver (tfra_table_step_prefetch_assign).
  python scripts/mb_overlap.py [--slots N] [--variants 0,1,2,4,8,9] [--out gpurun_out/mb_overlap.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--slots", type=int, default=999_999_992)
  ap.add_argument("--variants", default="0,1,2,4,8,9")
  ap.add_argument("--steps", type=int, default=64)
  ap.add_argument("--depth", default="1,4,8,16")
  ap.add_argument("--new-key-ratio", type=float, default=0.0)
  ap.add_argument("--dim", type=int, default=64)
  ap.add_argument("--dtype", default="float32")
  ap.add_argument("--skip-old", action="store_true")
  ap.add_argument("--fill-frac", type=float, default=1.0, help="pre-fill (and draw ids from) only this fraction of the ranks: below ~0.6 nothing is ever evicted, every id of a batch is resident")
  ap.add_argument("--ablate", default="0", help="comma list of TFRA_STEP_ABLATE masks (1 builders, 2 write-back + tail, 4 lookup return at once after 40 steps: timing only)")
  ap.add_argument("--one-ahead", action="store_true", help="D = 1: announce only the next batch (its plan is then built by a launch of its own)")
  ap.add_argument("--find-tiles", default="288", help="comma list of TFRA_STEP_OWN_SLICE (plan slots per write-back block)")
  ap.add_argument("--find-first", default="0", help="comma list of TFRA_STEP_FIND_FIRST (lookup blocks in front of the write-back's; -1: the driver's rule)")
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "mb_overlap.json"))
  args = ap.parse_args()
  import torch
  import tfra_amd.dynamic_embedding as de
  from bench import IdFactory, keys_of_ranks_torch, SEED, last_occurrence_rows
  dev = torch.device("cuda", 0)
  B, dim = 131072, args.dim
  dtype = getattr(torch, args.dtype)
  table = None
  for slots in [args.slots] + [int(args.slots * f) for f in (0.9, 0.8, 0.6, 0.4, 0.25)]:
    try:
      table = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots, max_capacity=slots, device=str(dev),
                              dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="mb_overlap")
      break
    except Exception as e:
      print("alloc failed at", slots, str(e)[:100], flush=True)
  gen = torch.Generator(device=dev).manual_seed(SEED)
  chunk = 4_000_000
  vals_fill = (torch.randn((chunk, dim), generator=gen, device=dev) * 0.01).to(dtype)
  t0 = time.perf_counter()
  n_ranks = int(slots * args.fill_frac)
  for lo in range(((n_ranks - 1) // chunk) * chunk + 1, 0, -chunk):   # coldest ranks first: the hot ids are the most recently used
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(n_ranks, lo + chunk - 1) + 1, dtype=torch.int64, device=dev))
    table._table.upsert(k, vals_fill[:k.numel()], unique_keys=True)
  resident = int(table.size().item())
  print("prefill %.1f s, resident %d of %d" % (time.perf_counter() - t0, resident, slots), flush=True)
  del vals_fill
  K = args.steps
  NB = 4 * K + 8
  idf = IdFactory(torch, dev, B, n_ranks, args.new_key_ratio, slots + 1, SEED + 7)
  ids = idf.keys(NB + 1)
  values = (torch.randn((B, dim), generator=gen, device=dev) * 0.01).to(dtype)
  outs = [torch.empty((B, dim), dtype=dtype, device=dev) for _ in range(8)]
  res = {"slots": slots, "resident": resident, "B": B, "dim": dim, "dtype": args.dtype, "new_key_ratio": args.new_key_ratio, "runs": []}

  def timed(fn, nwin=4):
    """fn(w) enqueues K steps of window w; returns (median us/step, host us/step)"""
    ts, hs = [], []
    for w in range(nwin):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      fn(w)
      h = time.perf_counter() - t0
      torch.cuda.synchronize()
      ts.append((time.perf_counter() - t0) / K * 1e6)
      hs.append(h / K * 1e6)
    o = sorted(range(nwin), key=lambda i: ts[i])[nwin // 2]
    return ts[o], hs[o], ts

  # ---- round-3 driver
  ps = de.PrefetchAssignStep(table).prime(ids[0])
  pos = [0]

  def old(w):
    for i in range(K):
      ps.step(values, ids[pos[0] + 1])
      pos[0] += 1

  if not args.skip_old:
    us, hus, all_ = timed(old)
    res["runs"].append({"driver": "step_prefetch_assign (round 3)", "us_per_step": us, "host_us_per_step": hus, "windows": all_})
    print(res["runs"][-1], flush=True)
  torch.cuda.synchronize()
  del ps
  depths = [int(x) for x in args.depth.split(",")]
  for v, ab, ft, ff in [(int(x), int(y), int(z), int(w)) for x in args.variants.split(",") for y in args.ablate.split(",")
                        for z in args.find_tiles.split(",") for w in args.find_first.split(",")]:
    os.environ["TFRA_STEP_VARIANT"] = str(v)
    os.environ["TFRA_STEP_ABLATE"] = str(ab)
    if ft > 0:
      os.environ["TFRA_STEP_OWN_SLICE"] = str(ft)
    else:
      os.environ.pop("TFRA_STEP_OWN_SLICE", None)   # 0: the driver's own (adaptive) choice
    os.environ["TFRA_STEP_FIND_FIRST"] = str(ff)
    for D in depths:
      drv = de.OverlapAssignStep(table)
      base = [0]
      if D == 1:
        drv.prime(ids[0])

        def one(w):
          for i in range(K):
            j = base[0]
            drv.step(values, ids[(j + 1) % NB], ids[(j + 2) % NB] if not args.one_ahead else None)
            base[0] += 1
        fn = one
      else:
        runs = []
        for w in range(4):
          rr = []
          for c in range(K // D):
            j0 = w * K + c * D
            il = [ids[(j0 + q) % NB] for q in range(D)]
            ol = [outs[q % 8] for q in range(D)]
            rr.append(drv.make_run(il, [values] * D, ol, ids_after=ids[(j0 + D) % NB], values_before=None if (w == 0 and c == 0) else values,
                                   ids_after2=ids[(j0 + D + 1) % NB]))
          runs.append(rr)

        def many(w):
          for r in runs[w]:
            r()
        fn = many
      tm = None
      if (v & 16) and D == 1:
        import numpy as np
        fn(0)
        drv.timing()     # re-arm: the stamps of the next 48 launches only
        torch.cuda.synchronize()
        K_save, K = K, 48
        fn(0)
        K = K_save
        sp = [[y if y is not None else (0.0, 0.0, 0.0, 0.0) for y in x] for x in drv.timing()]
        tm = {"launches": len(sp), "role_spans_us_median (start, end since the launch's first block)":
              {r: [float(np.median([x[i][j] for x in sp])) for j in range(4)] for i, r in enumerate(("build", "scatter", "write_back", "lookup", "tail", "map"))} if sp else None,
              "columns": "first block start, last block end, median block duration, p95 block duration"}
      us, hus, all_ = timed(fn)
      st = drv.stats()
      drv.flush()
      torch.cuda.synchronize()
      # last-occurrence-wins on the final batch
      jlast = ((base[0] - 1) if D == 1 else (4 * K - 1)) % NB
      got, ex = table.lookup(ids[jlast], return_exists=True)
      ok = bool(ex.all()) and bool(torch.equal(got, last_occurrence_rows(torch, ids[jlast], values)))
      if not ab:
        table._table.check_errors()
      res["runs"].append({"driver": "step_overlap", "variant": v, "ablate": ab, "own_slice": ft, "find_first": ff, "steps_per_host_call": D, "us_per_step": us, "host_us_per_step": hus,
                          "windows": all_, "stats": st, "last_batch_ok": ok, "timing": tm})
      print(res["runs"][-1], flush=True)
      del drv
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
  main()
