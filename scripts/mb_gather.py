"""gather_rows16_kernel variants: 131072 rows of 256 B from 22.5 K distinct rows (the position gather of the routed step), alone."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommenders-addons_amd"))
import torch
from tfra_amd import _capi
lib = _capi.lib()
dev = torch.device("cuda", 0)
B, U, dim = 131072, 22500, 64
rows = torch.randn((U, dim), device=dev)
outs = [torch.empty((B, dim), device=dev) for _ in range(4)]
idx = [torch.randint(0, U, (B,), device=dev, dtype=torch.int32) for _ in range(8)]
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(10):
  lib.tfra_gather_rows(B, dim * 4, P(rows), P(idx[i % 8]), P(outs[i % 4]), st)
e0.record()
for i in range(200):
  lib.tfra_gather_rows(B, dim * 4, P(rows), P(idx[i % 8]), P(outs[i % 4]), st)
e1.record(); torch.cuda.synchronize()
ok = bool(torch.equal(outs[3], rows[idx[(199) % 8].long()]))
print("variant %s: %.2f us per gather, correct %s" % (os.environ.get("TFRA_GATHER_VARIANT", "0"), e0.elapsed_time(e1) * 1e3 / 200, ok))
