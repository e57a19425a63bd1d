"""(e) on a box with at least TWO GPUs: the real RCCL transport (`tfra_rccl_transport_create`: grouped ncclSend / ncclRecv over xGMI on
the driver's own communicators) under `NativeRoutedStep`, one process per GPU, compared with ONE oracle table that sees every rank's
batches — exactly the check of tests/test_gpu_distributed.py (which stages the collectives through gloo because two ranks on one GPU
cannot form an RCCL communicator).  Skips on a 1-GPU box (the builder's pool); runs by itself wherever the suite meets 2 or 8 GPUs, so
the multi-rank RCCL path is exercised outside `bench.py --gpus N` too.

Reference: PY/shadow_embedding_ops.py:397-447 (__alltoall_embedding_lookup__), python/kernel_tests/horovod_sync_train_test.py:265-376."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

DIM, STEPS, LR = 8, 6, 0.5


def _n_gpus():
  try:
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0
  except Exception:   # noqa: BLE001
    return 0


def _batch(rank, step):
  rng = np.random.default_rng(7000 * step + rank)
  ids = (rng.zipf(1.3, size=(4, 600 + 70 * rank)).astype(np.int64) % 6000) * 7919 - 4321   # negative keys too
  g = (rng.standard_normal((ids.size, DIM)) * 0.01).astype(np.float32)
  return ids, g


def _worker(rank, world, port, out_dir):
  import torch
  import torch.distributed as dist
  import tfra_amd.dynamic_embedding as de
  from tfra_amd.dynamic_embedding.distributed import NativeRoutedStep
  os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  torch.cuda.set_device(rank)
  dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, device_id=torch.device("cuda", rank))
  try:
    dev = "cuda:%d" % rank
    opt = de.optimizers.SGD(LR)
    var = de.Variable(dim=DIM, name="rccl_w%d_r%d" % (world, rank), initializer=0.5, devices=[dev], **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    deo = de.DynamicEmbeddingOptimizer(opt)
    rs = NativeRoutedStep(var, deo, partition_mode=0, transport="rccl", max_batch=4096)
    assert rs.world == world and rs._rccl is not None          # the real transport: one RCCL communicator pair with `world` ranks
    batches = [_batch(rank, s) for s in range(STEPS)]
    dev_ids = [torch.from_numpy(b[0]).to(dev) for b in batches]
    torch.cuda.synchronize()
    for s in range(min(3, STEPS)):
      rs.feed(dev_ids[s])
    looked = []
    for step in range(STEPS):
      out = rs.lookup()
      looked.append(out.cpu().numpy().reshape(batches[step][0].shape + (DIM,)))
      rs.apply(torch.from_numpy(batches[step][1]).to(dev))
      if step + 3 < STEPS:
        rs.feed(dev_ids[step + 3])
    torch.cuda.synchronize()
    rs.close()
    k, v = var.export()
    k = k.cpu().numpy()
    assert np.all(((k & 0x7FFFFFFF) % world) == rank)           # default_partition_fn: this shard's keys only
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), keys=k, vals=v.cpu().numpy(), **{"look%d" % i: x for i, x in enumerate(looked)})
  finally:
    dist.destroy_process_group()


@pytest.mark.skipif(_n_gpus() < 2, reason="needs at least two GPUs on the box (one process per GPU, RCCL over xGMI)")
@pytest.mark.parametrize("world", [2, 8])
def test_native_route_over_real_rccl(world, tmp_path):
  import torch
  import torch.multiprocessing as mp
  import oracle
  from oracle import optimizers as oopt
  if torch.cuda.device_count() < world:
    pytest.skip("%d GPUs needed, %d visible" % (world, torch.cuda.device_count()))
  # (never wait forever for a collective that does not complete: a deadline, then the ranks are killed and the test fails)
  ctx = mp.spawn(_worker, args=(world, 29940 + world, str(tmp_path)), nprocs=world, join=False)
  import time
  deadline = time.monotonic() + 300.0
  done = False
  while not done and time.monotonic() < deadline:
    done = ctx.join(timeout=5.0)
  if not done:
    for pr in ctx.processes:
      if pr.is_alive():
        pr.kill()   # the exact processes started above
    pytest.fail("the %d-rank RCCL run did not finish within 300 s" % world)
  res = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
  tab = oracle.CpuTable(DIM)
  dflt = np.full(DIM, 0.5, np.float32)
  for step in range(STEPS):
    batches = [_batch(r, step) for r in range(world)]
    for r, (ids, g) in enumerate(batches):
      want = tab.find(ids.reshape(-1), dflt).reshape(ids.shape + (DIM,))
      np.testing.assert_allclose(res[r]["look%d" % step], want, rtol=1e-6, atol=1e-6)
    all_ids = np.concatenate([b[0].reshape(-1) for b in batches])
    all_g = np.concatenate([b[1] for b in batches])
    uniq, gsum, _ = oopt.segment_sum_by_key(all_ids, all_g)
    tab.insert(uniq, oopt.sgd(tab.find(uniq, dflt), gsum, LR))
  ek, ev = tab.export_sorted()
  gk = np.concatenate([r["keys"] for r in res])
  gv = np.concatenate([r["vals"] for r in res])
  o = np.argsort(gk)
  np.testing.assert_array_equal(gk[o], ek)            # every key lives on exactly one shard
  np.testing.assert_allclose(gv[o], ev, rtol=1e-6, atol=1e-6)


# ---- the metric's step on a sharded table (RoutedAssignStep, csrc/tfra_aroute.hip) over the REAL transport ------------------------------
AR_DIM, AR_STEPS = 16, 8


def _ar_universe(world):
  rng = np.random.default_rng(123)
  return rng.permutation(np.arange(1, 40001, dtype=np.int64)) * 104729 - 777_777


def _ar_batch(rank, step, universe):
  rng = np.random.default_rng(3000 * step + rank)
  n = 2500 + 100 * rank + (37 if step % 2 else 0)
  ids = universe[(rng.zipf(1.2, size=n) * 31 + rng.integers(0, 40, size=n)) % universe.size].astype(np.int64)
  ids[: n // 8] = universe[3]
  rng.shuffle(ids)
  vals = (np.arange(n, dtype=np.float32) + 10000.0 * (step + 1) + 5000.0 * rank)[:, None].repeat(AR_DIM, 1)
  return ids, vals


def _ar_worker(rank, world, port, out_dir):
  import torch
  import torch.distributed as dist
  import tfra_amd.dynamic_embedding as de
  from tfra_amd.dynamic_embedding.distributed import RoutedAssignStep
  os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
  torch.cuda.set_device(rank)
  dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, device_id=torch.device("cuda", rank))
  try:
    dev = "cuda:%d" % rank
    universe = _ar_universe(world)
    t = de.CuckooHashTable(torch.int64, torch.float32, torch.zeros(AR_DIM), device=dev, dim=AR_DIM, name="ar_rccl_%d" % rank)
    mine = universe[((universe & 0x7FFFFFFF) % world) == rank][::2]      # half of this shard's keys resident, the rest enter through the steps
    k = torch.from_numpy(mine).to(dev)
    t._table.upsert(k, (k % 1000).to(torch.float32)[:, None].repeat(1, AR_DIM), unique_keys=True)
    rs = RoutedAssignStep(t, transport="rccl", max_batch=1 << 14)
    assert rs.rccl_ranks == world and not rs.identity       # the real transport: RCCL itself reports `world` ranks per communicator
    batches = [_ar_batch(rank, s, universe) for s in range(AR_STEPS)]
    ids_t = [torch.from_numpy(b[0]).to(dev) for b in batches]
    vals_t = [torch.from_numpy(b[1]).to(dev) for b in batches]
    torch.cuda.synchronize()
    fed = 0
    for _ in range(4):
      rs.feed(ids_t[fed]); fed += 1
    looked = []
    for s in range(AR_STEPS):
      rows = rs.step(vals_t[s - 1] if s else None)
      if fed < AR_STEPS:
        rs.feed(ids_t[fed]); fed += 1
      looked.append(rows.cpu().numpy())
    rs.flush(vals_t[-1])
    torch.cuda.synchronize()
    t._table.check_errors()
    kk, vv = t.export()
    kk = kk.cpu().numpy()
    assert np.all(((kk & 0x7FFFFFFF) % world) == rank)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), keys=kk, vals=vv.cpu().numpy(), **{"look%d" % i: x for i, x in enumerate(looked)})
    rs.close()
  finally:
    dist.destroy_process_group()


def _ar_check(world, tmp_path):
  import oracle
  res = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
  universe = _ar_universe(world)
  tab = oracle.CpuTable(AR_DIM)
  pre = np.concatenate([universe[((universe & 0x7FFFFFFF) % world) == r][::2] for r in range(world)])
  tab.insert(pre, (pre % 1000).astype(np.float32)[:, None].repeat(AR_DIM, 1))
  dflt = np.zeros(AR_DIM, np.float32)
  for step in range(AR_STEPS):
    batches = [_ar_batch(r, step, universe) for r in range(world)]
    for r, (ids, _) in enumerate(batches):            # every rank looks up first ...
      np.testing.assert_array_equal(res[r]["look%d" % step], tab.find(ids, dflt), err_msg="rank %d step %d" % (r, step))
    for ids, vals in batches:                         # ... then rank 0's insert_or_assign, rank 1's, ...: the last occurrence wins
      tab.insert(ids, vals)
  ek, ev = tab.export_sorted()
  gk = np.concatenate([r["keys"] for r in res])
  gv = np.concatenate([r["vals"] for r in res])
  o = np.argsort(gk)
  np.testing.assert_array_equal(gk[o], ek)
  np.testing.assert_array_equal(gv[o], ev)


def _ar_run(world, port, tmp_path):
  import time
  import torch.multiprocessing as mp
  ctx = mp.spawn(_ar_worker, args=(world, port, str(tmp_path)), nprocs=world, join=False)
  deadline = time.monotonic() + 300.0
  done = False
  while not done and time.monotonic() < deadline:
    done = ctx.join(timeout=5.0)
  if not done:
    for pr in ctx.processes:
      if pr.is_alive():
        pr.kill()   # the exact processes started above
    pytest.fail("the %d-rank RCCL run did not finish within 300 s" % world)


def test_routed_assign_over_real_rccl_one_rank(tmp_path):
  """Runs on ANY box with a GPU: the routed assign step over the REAL RCCL transport with a one-rank communicator pair — every
  alltoall is a grouped ncclSend / ncclRecv to the rank itself, on the driver's own communicators (tfra_rccl_transport_create) —
  against the oracle table.  What a 1-GPU box can exercise of the multi-rank path: the transport's code, its two channels, the
  event order between the driver's stream and the caller's."""
  _ar_run(1, 29951, tmp_path)
  _ar_check(1, tmp_path)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs at least two GPUs on the box (one process per GPU, RCCL over xGMI)")
@pytest.mark.parametrize("world", [2, 8])
def test_routed_assign_over_real_rccl(world, tmp_path):
  import torch
  if torch.cuda.device_count() < world:
    pytest.skip("%d GPUs needed, %d visible" % (world, torch.cuda.device_count()))
  _ar_run(world, 29952 + world, tmp_path)
  _ar_check(world, tmp_path)


def test_rccl_entry_points_exist_and_refuse_bad_arguments():
  """On any box: the transport's entry points are exported and fail loudly (no GPU pair needed)."""
  import ctypes
  from tfra_amd import _capi
  lib = _capi.lib()
  assert hasattr(lib, "tfra_rccl_unique_id") and hasattr(lib, "tfra_rccl_transport_create") and hasattr(lib, "tfra_rccl_transport_destroy")
  tr = _capi.Transport()
  rc = lib.tfra_rccl_transport_create(b"/nonexistent/librccl.so", b"\0" * (2 * _capi.RCCL_ID_BYTES), 0, 2, 0, ctypes.byref(tr))
  assert rc != 0 and b"rccl" in lib.tfra_last_error().lower()
