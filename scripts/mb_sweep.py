"""Does upsert_own_kernel run in 'rounds' of resident waves?  Direct path (tfra_table_insert_or_assign, UNIQUE) on the 10^9-slot
table with n unique keys of configs[2]-type batches (84 % never-seen), n swept across the resident capacity (4 waves/SIMD x 1024
SIMDs x 16 keys = 65 536 keys).  Run under rocprofv3 --kernel-trace; the per-dispatch durations are read from the trace."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  sys.path.insert(0, p)
import numpy as np, torch
import tfra_amd.dynamic_embedding as de
from tfra_amd import _capi
from bench import keys_of_ranks_torch, keys_of_ranks, mixed_batches

dev = torch.device("cuda", 0)
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
B, dim, dtype = 131072, 64, torch.float32
t = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots, max_capacity=slots, device="cuda:0",
                    dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="mb_sweep")
vals = torch.randn((4_000_000, dim), device=dev) * 0.01
for lo in range(1, slots + 1, 4_000_000):
  k = keys_of_ranks_torch(torch, torch.arange(lo, min(slots, lo + 3_999_999) + 1, dtype=torch.int64, device=dev))
  t._table.upsert(k, vals[:k.numel()], unique_keys=True)
torch.cuda.synchronize()
tbl = t._table
lib = _capi.lib()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda x: ctypes.c_void_p(x.data_ptr())
rng = np.random.default_rng(11)
NS = [int(x) for x in os.environ.get("SWEEP", "8192,16384,32768,49152,57344,63488,65536,67584,73728,78000,98304,131072").split(",")]
REP = 5
nb = len(NS) * REP
# enough unique keys per launch: two batches' worth when n > one batch's unique count
ranks, _ = mixed_batches(rng, 2 * nb, B, slots, ratio, slots + 1)
ids = torch.from_numpy(keys_of_ranks(ranks.reshape(-1)).reshape(2 * nb, B)).to(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
order = []
res = {}
j = 0
for n in NS:
  ts = []
  for r in range(REP):
    u = torch.unique(torch.cat([ids[2 * j], ids[2 * j + 1]]))[:n].contiguous()
    j += 1
    perm = torch.randperm(u.numel(), device=dev)
    u = u[perm].contiguous()          # not sorted by key value
    torch.cuda.synchronize()
    e0.record()
    _capi.check(lib.tfra_table_insert_or_assign(tbl._h, u.numel(), P(u), P(vals), None, 1, st))
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
    order.append(int(u.numel()))
  res[n] = [round(x, 1) for x in ts]
tbl.check_errors()
print(json.dumps({"ratio": ratio, "event_us_per_call": res, "order": order}))
json.dump({"ratio": ratio, "event_us_per_call": res, "order": order}, open(os.path.join(ROOT, "gpurun_out", "mb_sweep_%s.json" % sys.argv[2] if len(sys.argv) > 2 else "x"), "w"))
