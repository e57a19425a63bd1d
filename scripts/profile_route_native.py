"""Host time of each call of the C routed step (tfra_route_feed / _lookup / _apply), one rank, RCCL world of 1, c2 shape.
  python scripts/profile_route_native.py [transport: rccl|none]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import tfra_amd.dynamic_embedding as de
from tfra_amd.dynamic_embedding.distributed import NativeRoutedStep
from bench import keys_of_ranks, keys_of_ranks_torch, zipf_bounded

mode = sys.argv[1] if len(sys.argv) > 1 else "rccl"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29552")
torch.cuda.set_device(0)
if mode == "rccl":
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
N, B, DIM = 20_000_000, 131072, 64
opt = de.optimizers.Adam(1e-3)
deo = de.DynamicEmbeddingOptimizer(opt)
var = de.Variable(dim=DIM, name="prof", initializer=0.0, init_size=int(N * 1.05), devices=["cuda:0"], **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
for lo in range(1, N + 1, 4_000_000):
  k = keys_of_ranks_torch(torch, torch.arange(lo, lo + 4_000_000, dtype=torch.int64, device="cuda"))
  var.tables[0]._table.upsert(k, torch.zeros((k.numel(), DIM), device="cuda"), unique_keys=True)
rng = np.random.default_rng(0)
ids = [torch.from_numpy(keys_of_ranks(zipf_bounded(rng, B, N))).cuda() for _ in range(8)]
g = torch.randn((B, DIM), device="cuda") * 0.01
AHEAD = int(os.environ.get("TFRA_ROUTE_AHEAD", "3"))
rs = NativeRoutedStep(var, deo, force_collectives=True, max_batch=B, threaded=os.environ.get("TFRA_ROUTE_THREAD", "1") != "0")
for j in range(AHEAD):
  rs.feed(ids[j])
T = {"lookup": 0.0, "apply": 0.0, "feed": 0.0}


def step(i, sync=False):
  for name, fn in (("lookup", rs.lookup), ("apply", lambda: rs.apply(g)), ("feed", lambda: rs.feed(ids[(i + AHEAD) & 7]))):
    t0 = time.perf_counter()
    fn()
    if sync:
      torch.cuda.synchronize()
    T[name] += time.perf_counter() - t0


for i in range(20):
  step(i)
torch.cuda.synchronize()
for sync in (False, True):
  for k in T:
    T[k] = 0.0
  t0 = time.perf_counter()
  for i in range(200):
    step(20 + i, sync)
  torch.cuda.synchronize()
  tot = (time.perf_counter() - t0) * 1e6 / 200
  print("sync after each call" if sync else "free running", "us per step %.1f" % tot, {k: round(v * 1e6 / 200, 1) for k, v in T.items()})
for _ in range(AHEAD):
  rs.lookup(); rs.apply(g)
torch.cuda.synchronize()
rs.close()
if mode == "rccl":
  dist.destroy_process_group()
