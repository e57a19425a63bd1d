"""Where does the HOST time of the routed multi-GPU step go?  One rank, collectives forced on (RCCL world of 1), cProfile
over 200 steps of RoutedPrefetchStep at the c2 shape.   python scripts/profile_route_host.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import tfra_amd.dynamic_embedding as de
from tfra_amd.dynamic_embedding.distributed import RoutedPrefetchStep
from bench import keys_of_ranks, keys_of_ranks_torch, zipf_bounded

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29551")
torch.cuda.set_device(0)
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
rs = RoutedPrefetchStep(var, deo, force_collectives=True)
rs.feed(ids[0]); rs.feed(ids[1])


def step(i):
  rs.lookup(); rs.apply(g); rs.feed(ids[(i + 2) & 7])


for i in range(20):
  step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(200):
  step(20 + i)
pr.disable()
torch.cuda.synchronize()
print("us per step (wall): %.1f" % ((time.perf_counter() - t0) * 1e6 / 200))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
dist.destroy_process_group()
