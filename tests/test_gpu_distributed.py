"""A15 / (e) with world_size 2 ON THE DEVICE: two processes share cuda:0, each owns the HIP table of its key-hash shard
and runs the real front-end kernels (tfra_unique / tfra_partition / tfra_reduce_by_key / gather / scatter); the
collectives go through gloo, staged over the host (RCCL needs one GPU per rank; the N>1 route itself — split sizes,
owner-major order, the way back, the gradient route — is the same code).  Checked against ONE oracle table that sees
the union of both ranks' batches: lookups bit-exact, the SGD write-back to <= 1e-6.

Reference: PY/shadow_embedding_ops.py:397-447 (__alltoall_embedding_lookup__), python/kernel_tests/
horovod_sync_train_test.py:265-376 (sharded training equals the single-table result)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

DIM, STEPS, LR = 8, 5, 0.5


def _batch(rank, step):
  rng = np.random.default_rng(1000 * step + rank)
  ids = (rng.zipf(1.3, size=(4, 700 + 90 * rank)).astype(np.int64) % 5000) * 7919 - 12345   # negative keys too
  g = (rng.standard_normal((ids.size, DIM)) * 0.01).astype(np.float32)   # synthetic gradients N(0, 1e-2)
  return ids, g


def _worker(rank, world, port, dedup, out_dir):
  if dedup >= 2:
    return _worker_prefetched(rank, world, port, out_dir, native=dedup == 3)
  import torch
  import torch.distributed as dist
  import tfra_amd.dynamic_embedding as de
  from tfra_amd.dynamic_embedding.distributed import AllToAllEmbedding
  os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
  torch.cuda.set_device(0)
  dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
  try:
    opt = de.optimizers.SGD(LR)
    var = de.Variable(dim=DIM, name="a2a_w2_r%d_%d" % (rank, dedup), initializer=0.5, devices=["cuda:0"],
                      **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    deo = de.DynamicEmbeddingOptimizer(opt)
    emb = AllToAllEmbedding(var, partition_mode=0, dedup=bool(dedup))
    assert emb.world == 2 and not emb.passthrough
    looked = []
    for step in range(STEPS):
      ids, g = _batch(rank, step)
      out = emb.lookup(torch.from_numpy(ids).cuda())
      looked.append(out.cpu().numpy())
      emb.apply_gradients(deo, torch.from_numpy(g).cuda())
    k, v = var.export()
    k = k.cpu().numpy()
    assert np.all(((k & 0x7FFFFFFF) % world) == rank)   # default_partition_fn, CUDA-build branch: this shard's keys only
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), keys=k, vals=v.cpu().numpy(), **{"look%d" % i: x for i, x in enumerate(looked)})
  finally:
    dist.destroy_process_group()


def _worker_prefetched(rank, world, port, out_dir, native=False):
  """The same training run through RoutedPrefetchStep: the id-only half of the route two batches ahead on a second stream.
  native: through the C driver (tfra_route_*) with the host-staged transport (RCCL cannot pair two ranks on one GPU)."""
  import torch
  import torch.distributed as dist
  import tfra_amd.dynamic_embedding as de
  from tfra_amd.dynamic_embedding.distributed import NativeRoutedStep, RoutedPrefetchStep
  os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
  torch.cuda.set_device(0)
  dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
  try:
    opt = de.optimizers.SGD(LR)
    var = de.Variable(dim=DIM, name="a2a_w2p_r%d" % rank, initializer=0.5, devices=["cuda:0"],
                      **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    deo = de.DynamicEmbeddingOptimizer(opt)
    if native:
      rs = NativeRoutedStep(var, deo, partition_mode=0, transport="staged", max_batch=4096)
      assert rs.world == 2
    else:
      rs = RoutedPrefetchStep(var, deo, partition_mode=0)
      assert rs.world == 2 and rs.collectives
    batches = [_batch(rank, s) for s in range(STEPS)]
    dev_ids = [torch.from_numpy(b[0]).cuda() for b in batches]
    torch.cuda.synchronize()
    for s in range(min(2, STEPS)):
      rs.feed(dev_ids[s])
    looked = []
    for step in range(STEPS):
      out = rs.lookup()
      looked.append(out.cpu().numpy().reshape(batches[step][0].shape + (DIM,)))
      rs.apply(torch.from_numpy(batches[step][1]).cuda())
      if step + 2 < STEPS:
        rs.feed(dev_ids[step + 2])
    assert deo.iterations == STEPS
    if native:
      torch.cuda.synchronize()
      rs.close()
    k, v = var.export()
    k = k.cpu().numpy()
    assert np.all(((k & 0x7FFFFFFF) % world) == rank)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), keys=k, vals=v.cpu().numpy(), **{"look%d" % i: x for i, x in enumerate(looked)})
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("dedup", [1, 0, 2, 3])   # 2 = RoutedPrefetchStep (route prepared two batches ahead), 3 = the C driver
def test_alltoall_world2_real_tables_one_gpu(dedup, tmp_path):
  import torch
  import torch.multiprocessing as mp
  import oracle
  from oracle import optimizers as oopt
  assert torch.cuda.is_available()
  world, port = 2, 29960 + dedup
  mp.spawn(_worker, args=(world, port, dedup, str(tmp_path)), nprocs=world, join=True)
  res = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
  # the oracle: one table, per step every rank looks up first (synchronous training), then all gradients are applied
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


@pytest.mark.parametrize("route", ["native", "prefetch"])
def test_bench_two_ranks_one_gpu(route, tmp_path):
  """The driver's N > 1 invocation of bench.py (torch.distributed.run, one process per rank) in small: two ranks share
  cuda:0, collectives host-staged through gloo (RCCL cannot pair two ranks on one GPU).  One JSON line from rank 0 with
  the contract's fields; throughput > 0."""
  import json
  import subprocess
  import sys
  env = dict(os.environ, TFRA_BENCH_BACKEND="gloo", TFRA_BENCH_ROUTE=route, HSA_ENABLE_IPC_MODE_LEGACY="0", TFRA_BENCH_DETAIL_DIR=str(tmp_path))
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(29990 + (route == "native")), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
         "--config", "c4", "--c4-keys", "300000", "--batch", "8192", "--no-cpu-baseline"]
  p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
  assert p.returncode == 0, p.stderr[-3000:]
  lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, p.stdout[-2000:]
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0
  assert d["config"]["global_batch"] == 2 * 8192
  assert "roofline" in d and d["unit"] == "pairs/s"
  # `--config c4`: BASELINE configs[3] with the gradient route (hash-sharded, per-GPU batch from the global Zipf, fused SGD at the owner)
  assert d["config"]["workload"].startswith("BASELINE configs[3]") and d["config"]["route"] == route
  assert len(lines[0]) < 6000          # the driver keeps an ~8 KB tail of stdout: the line must fit whole
  full = json.load(open(os.path.join(str(tmp_path), "bench_detail.json")))   # everything else: the detail file
  assert full["config"]["timing"]["value"]["windows"] == 5 and full["config"]["timing"]["value"]["steps_per_window"] == 6


def test_bench_gpus2_without_a_launcher_runs_the_metric_step_sharded(tmp_path):
  """`python bench.py --gpus 2` exactly as the driver starts it (NO torch.distributed.run in front): bench.py starts its own two ranks
  (here sharing cuda:0, collectives host-staged through gloo) and measures, per GPU, the METRIC's workload — lookup(B) +
  insert_or_assign(B) on a bounded LRU shard at capacity, ids / rows / values routed, the owner running the overlapped step — the
  same step the one-GPU line reports.  One JSON line from rank 0, its in-run verification true."""
  import json
  import subprocess
  import sys
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
  env.update(TFRA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", TFRA_BENCH_DETAIL_DIR=str(tmp_path))
  cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--shard-slots", "300000",
         "--batch", "8192", "--no-cpu-baseline"]
  p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
  assert p.returncode == 0, p.stderr[-3000:]
  lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, p.stdout[-2000:]
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["unit"] == "pairs/s"
  assert d["config"]["global_batch"] == 2 * 8192 and d["config"]["route"] == "assign_route" and d["config"]["driver"] == "routed_overlapped_step"
  assert d["config"]["workload"].startswith("BASELINE metric") and "lookup+insert pairs/s" in d["metric"]
  assert d["verified"] is True and d["roofline"]["step_frac"] > 0 and d["roofline"]["avg_launch_us"] > 0
  assert len(lines[0]) < 6000
  full = json.load(open(os.path.join(str(tmp_path), "bench_detail.json")))
  st = full["config"]["route_stats"]
  assert st["owner_overlapped"] >= 5 * 6 and full["config"]["verified"]["routed_step_last_batch"] is True


def test_bench_m1s_one_rank_through_the_route_driver(tmp_path):
  """`python bench.py --config m1s` (one rank): the per-GPU workload of every `--gpus N` run THROUGH the route driver, device buffers
  aliased where the alltoalls would be — the line's `scaling_point` on a table of its own."""
  import json
  import subprocess
  import sys
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
  env.update(TFRA_BENCH_DETAIL_DIR=str(tmp_path))
  cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "m1s", "--steps", "6", "--warmup", "2", "--shard-slots", "400000",
         "--batch", "8192", "--no-cpu-baseline"]
  p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
  assert p.returncode == 0, p.stderr[-3000:]
  d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
  assert d["n_gpus"] == 1 and d["verified"] is True and d["config"]["route"] == "assign_route" and d["value"] > 0
  assert "no transport" in d["config"]["parallelism"] and d["roofline"]["avg_launch_us"] > 0


# ---- configs[4] across ranks: several tables, each hash-sharded, one route per table on ONE shared transport, fused FTRL ----------
MT_DIMS, MT_STEPS = (8, 16, 4), 4


def _mt_batch(rank, step, table):
  rng = np.random.default_rng(9000 * step + 10 * rank + table)
  ids = (rng.zipf(1.3, size=900 + 60 * rank + 30 * table).astype(np.int64) % 4000) * 104729 + table * 7 - 999
  g = (rng.standard_normal((ids.size, MT_DIMS[table])) * 0.01).astype(np.float32)
  return ids, g


def _worker_multi_table(rank, world, port, out_dir):
  import torch
  import torch.distributed as dist
  import tfra_amd.dynamic_embedding as de
  from tfra_amd.dynamic_embedding.distributed import MultiTableRoutedStep
  os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
  torch.cuda.set_device(0)
  dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
  try:
    opt = de.optimizers.Ftrl(0.05, l1_regularization_strength=1e-3, l2_regularization_strength=1e-3)
    deo = de.DynamicEmbeddingOptimizer(opt)
    tabs = [de.Variable(dim=d, name="mt_w2_t%d_r%d" % (i, rank), initializer=0.25, devices=["cuda:0"], **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
            for i, d in enumerate(MT_DIMS)]
    ms = MultiTableRoutedStep(tabs, deo, partition_mode=0, transport="staged", max_batch=4096)
    assert ms.world == 2 and len(ms.steps) == len(MT_DIMS)
    assert all(s_._staged is ms.steps[0]._staged for s_ in ms.steps)      # ONE transport for all tables
    batches = [[_mt_batch(rank, s, t) for t in range(len(MT_DIMS))] for s in range(MT_STEPS)]
    dev_ids = [[torch.from_numpy(b[0]).cuda() for b in row] for row in batches]
    torch.cuda.synchronize()
    for s in range(min(2, MT_STEPS)):
      ms.feed(dev_ids[s])
    looked = []
    for step in range(MT_STEPS):
      rows = ms.lookup()
      looked.append([r.cpu().numpy() for r in rows])
      ms.apply([torch.from_numpy(b[1]).cuda() for b in batches[step]])
      if step + 2 < MT_STEPS:
        ms.feed(dev_ids[step + 2])
    assert deo.iterations == MT_STEPS          # ONE optimizer step per multi-table apply
    torch.cuda.synchronize()
    ms.close()
    out = {}
    for i, v in enumerate(tabs):
      k, val = v.export()
      k = k.cpu().numpy()
      assert np.all(((k & 0x7FFFFFFF) % world) == rank)
      out["keys%d" % i], out["vals%d" % i] = k, val.cpu().numpy()
      for s in range(MT_STEPS):
        out["look%d_%d" % (s, i)] = looked[s][i]
    np.savez(os.path.join(out_dir, "mt_rank%d.npz" % rank), **out)
  finally:
    dist.destroy_process_group()


def test_multi_table_route_world2_ftrl_vs_oracle(tmp_path):
  """BASELINE configs[4] in small: three tables of different dims, each sharded over two ranks (both on cuda:0, collectives staged
  through gloo), MultiTableRoutedStep with fused FTRL — every table equal to ONE oracle table that sees both ranks' batches, rule by
  oracle/optimizers.py (pinned to TensorFlow's published FTRL answers)."""
  import torch
  import torch.multiprocessing as mp
  import oracle
  from oracle import optimizers as oopt
  assert torch.cuda.is_available()
  world = 2
  mp.spawn(_worker_multi_table, args=(world, 29975, str(tmp_path)), nprocs=world, join=True)
  res = [np.load(tmp_path / ("mt_rank%d.npz" % r)) for r in range(world)]
  hp = dict(lr=0.05, lr_power=-0.5, l1=1e-3, l2=1e-3, init_acc=0.1)
  for t, dim in enumerate(MT_DIMS):
    tabs = [oracle.CpuTable(dim) for _ in range(3)]
    ora = oopt.SparseOptimizerOracle("ftrl", tabs[0], tabs[1:], hp, 0.25)
    dflt = np.full(dim, 0.25, np.float32)
    for step in range(MT_STEPS):
      batches = [_mt_batch(r, step, t) for r in range(world)]
      for r, (ids, g) in enumerate(batches):
        np.testing.assert_allclose(res[r]["look%d_%d" % (step, t)], tabs[0].find(ids, dflt), rtol=1e-6, atol=1e-6)
      ora.apply(np.concatenate([b[0] for b in batches]), np.concatenate([b[1] for b in batches]))
    ek, ev = tabs[0].export_sorted()
    gk = np.concatenate([r["keys%d" % t] for r in res])
    gv = np.concatenate([r["vals%d" % t] for r in res])
    o = np.argsort(gk)
    np.testing.assert_array_equal(gk[o], ek)
    np.testing.assert_allclose(gv[o], ev, rtol=2e-6, atol=2e-6)
