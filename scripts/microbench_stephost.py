"""Where does the host time of one PrefetchStep.step go?  (Python around the call vs the C call itself)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import bench
import tfra_amd.dynamic_embedding as de
from tfra_amd import _capi

B, DIM, N = 131072, 64, 10_000_000
opt = de.optimizers.Adam(1e-3)
deo = de.DynamicEmbeddingOptimizer(opt)
var = de.Variable(dim=DIM, name="sh", initializer=0.0, init_size=int(N * 1.05), **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
for lo in range(0, N, 2_000_000):
  k = bench.keys_of_ranks_torch(torch, torch.arange(lo + 1, lo + 2_000_001, device="cuda"))
  var.upsert(k, torch.zeros((k.numel(), DIM), device="cuda"))
rng = np.random.default_rng(0)
ids = [torch.from_numpy(bench.keys_of_ranks(bench.zipf_bounded(rng, B, N))).cuda() for _ in range(8)]
g = torch.randn((B, DIM), device="cuda") * 0.01
ps = de.PrefetchStep(var, deo).prime(ids[0])
tc = [0.0]
orig = _capi.call
def timed_call(name, *a):
  t0 = time.perf_counter(); r = orig(name, *a); tc[0] += time.perf_counter() - t0; return r
_capi.call = timed_call
import tfra_amd.dynamic_embedding.optimizer as om
om._capi.call = timed_call
for i in range(20):
  ps.step(g, ids[(i + 1) & 7])
torch.cuda.synchronize()
tc[0] = 0.0
R = 300
t0 = time.perf_counter()
for i in range(R):
  ps.step(g, ids[(i + 1) & 7])
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per step: wall %.1f us, host enqueue %.1f us, of which inside the C call %.1f us" % (t_all / R * 1e6, t_host / R * 1e6, tc[0] / R * 1e6))
