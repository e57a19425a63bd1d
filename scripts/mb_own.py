"""Write-back variants on the 10^9-slot table (round 3): per process ONE setting of TFRA_OWN_FINISH / TFRA_OWN_NT (they are
read once), measured on the metric's shape (dim 64 fp32, Zipf-1.2 resident ids: m1b) and on configs[2]'s batch shape
(50 % never-seen ids: c3; same 256-B rows).  Prints one JSON line; run under gpurun for each variant.
  python scripts/mb_own.py [slots] [tag]"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  sys.path.insert(0, p)
import numpy as np, torch
import tfra_amd.dynamic_embedding as de
from tfra_amd import _capi
from bench import keys_of_ranks_torch, keys_of_ranks, zipf_bounded, mixed_batches, raw_calls, Timer, SEED

dev = torch.device("cuda", 0)
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
tag = sys.argv[2] if len(sys.argv) > 2 else ""
B, dim, dtype = 131072, 64, torch.float32
t = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots, max_capacity=slots, device="cuda:0",
                    dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="mb_own")
vals = torch.randn((4_000_000, dim), device=dev) * 0.01
for lo in range(1, slots + 1, 4_000_000):
  k = keys_of_ranks_torch(torch, torch.arange(lo, min(slots, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
  t._table.upsert(k, vals[:k.numel()], unique_keys=True)
torch.cuda.synchronize()
tbl = t._table
lib = _capi.lib()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda x: ctypes.c_void_p(x.data_ptr())
tm = Timer(torch)
rc = raw_calls(torch, dev)
QUICK = os.environ.get("MB_QUICK") == "1"
rep = {"tag": tag, "ablate": os.environ.get("TFRA_OWN_ABLATE", "0"), "finish": os.environ.get("TFRA_OWN_FINISH", "inside"), "nt": os.environ.get("TFRA_OWN_NT", "512"), "slots": slots,
       "size": int(t.size().item())}
rng = np.random.default_rng(3)
NB = 48
fresh = slots + 1
WORK = os.environ.get("MB_WORK", "m1b,c3").split(",")
for name, ratio in (("m1b", 0.0), ("c3", 0.5)):
  if name not in WORK:
    continue
  ranks, fresh = mixed_batches(rng, NB, B, slots, ratio, fresh)
  ids = torch.from_numpy(keys_of_ranks(ranks.reshape(-1)).reshape(NB, B)).to(dev)
  out = torch.empty((B, dim), dtype=dtype, device=dev)
  v1 = vals[:B]
  r = {}
  finds = [rc.find(tbl._h, ids[j], out, tbl._default_value) for j in range(8)]
  r["find_us"] = tm.us(lambda i: finds[i % 8](), reps=16, warm=3)
  # planned write-back (plan built before), a different batch per launch
  plans = [de.table_ops.SparsePlan(dev, 0) for _ in range(12)]
  for j, pl in enumerate(plans):
    pl.build(ids[8 + j], sync=False)
  torch.cuda.synchronize()
  c = plans[0].read()[0]
  r["U"] = c["many"] + c["few"]
  ups = [rc.upsert_planned(tbl._h, plans[j], v1) for j in range(12)]
  r["upsert_planned_us"] = tm.us(lambda i: ups[i](), reps=9, warm=3)
  # direct: the unique keys of a batch through tfra_table_insert_or_assign(UNIQUE)
  uq = [torch.unique(ids[20 + j]) for j in range(12)]
  ins = [(lambda a=(tbl._h, uq[j].numel(), P(uq[j]), P(v1), None, 1, st): _capi.check(lib.tfra_table_insert_or_assign(*a))) for j in range(12)]
  r["insert_unique_us"] = tm.us(lambda i: ins[i](), reps=9, warm=3)
  if QUICK:
    rep[name] = r
    continue
  tbl.check_errors()
  # parity spot check of the direct path: the rows of the last call are what a lookup returns
  got = tbl.find(uq[11])
  r["insert_unique_rows_ok"] = bool(torch.equal(got, v1[:uq[11].numel()]))
  # planned path: the row of a key = the value row of its LAST occurrence
  b = ids[8 + 11]
  got = tbl.find(b)
  order = torch.arange(B, device=dev)
  uk, inv = torch.unique(b, return_inverse=True)
  lp = torch.zeros(uk.numel(), dtype=torch.long, device=dev)
  lp.scatter_reduce_(0, inv, order, reduce="amax", include_self=False)
  exp = v1[lp][inv]
  # batches 8+11 was written before batches 20.. (direct) which may have overwritten shared hot keys: compare only keys not in uq
  later = torch.cat([u for u in uq])
  mask = ~torch.isin(b, later)
  r["planned_rows_checked"] = int(mask.sum().item())
  r["planned_rows_ok"] = bool(torch.equal(got[mask], exp[mask]))
  # steps: fresh batches for every step of every driver (never-seen ids stay never-seen)
  K, W = 100, 10
  sranks, fresh = mixed_batches(rng, 3 * (K + W + 2), B, slots, ratio, fresh)
  sids = torch.from_numpy(keys_of_ranks(sranks.reshape(-1)).reshape(3 * (K + W + 2), B)).to(dev)
  ps = de.PrefetchAssignStep(t).prime(sids[0])
  seq = [sids[i] for i in range(K + W + 2)]
  for i in range(W):
    ps.step(v1, seq[i + 1])
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(K):
    ps.step(v1, seq[W + i + 1])
  torch.cuda.synchronize(); r["step_prefetch_us"] = (time.perf_counter() - t0) / K * 1e6
  seq = [sids[K + W + 2 + i] for i in range(K + W + 2)]
  def plain(i):
    tbl.find(seq[i]); tbl.upsert_sparse(seq[i], v1)
  for i in range(W): plain(i)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(K): plain(W + i)
  torch.cuda.synchronize(); r["step_plain_us"] = (time.perf_counter() - t0) / K * 1e6
  # op surface, raw C calls: find(B) ; insert_or_assign(U unique keys, UNIQUE)  (tf.unique is TF's, precomputed here)
  seq = [sids[2 * (K + W + 2) + i] for i in range(K + W)]
  fu = [rc.find(tbl._h, x, out, tbl._default_value) for x in seq]
  uq2 = [torch.unique(x) for x in seq]
  iu = [(lambda a=(tbl._h, u.numel(), P(u), P(v1), None, 1, st): _capi.check(lib.tfra_table_insert_or_assign(*a))) for u in uq2]
  def ops(i):
    fu[i](); iu[i]()
  for i in range(W): ops(i)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(K): ops(W + i)
  torch.cuda.synchronize(); r["step_find_insert_unique_us"] = (time.perf_counter() - t0) / K * 1e6
  r["tfra_unique_us"] = tm.us(lambda i: de.device_ops.unique_no_sync(ids[i % NB]), reps=16, warm=3)
  tbl.check_errors()
  r["census"] = {k: int(v) for k, v in tbl.slot_census().items()} if name == "c3" else None
  r["size"] = int(t.size().item())
  rep[name] = r
print(json.dumps(rep), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "mb_own.jsonl"), "a") as f:
  f.write(json.dumps(rep) + "\n")
