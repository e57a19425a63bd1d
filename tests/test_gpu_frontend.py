"""GPU parity: front-end device ops, sharded Variable, embedding_lookup*, fused optimizers — all
through the C ABI — against the numpy restatements in oracle/."""
import numpy as np
import pytest

import oracle
from oracle import frontends as ofe
from oracle import optimizers as oopt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  import tfra_amd.dynamic_embedding as de
  return torch, de


def T(torch, a):
  return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n,hi", [(1, 5), (7, 3), (1000, 50), (131072, 20000), (300001, 10**12)])
def test_unique_is_tf_unique(env, n, hi):
  torch, de = env
  rng = np.random.default_rng(n)
  ids = rng.integers(-hi, hi, size=n).astype(np.int64)
  ids[0] = np.iinfo(np.int64).min  # the scratch set's sentinel value is a legal id
  u, idx, cnt = de.device_ops.unique(T(torch, ids))
  eu, eidx = ofe.unique(ids)
  assert int(cnt.item()) == eu.size
  np.testing.assert_array_equal(u.cpu().numpy(), eu)
  np.testing.assert_array_equal(idx.cpu().numpy(), eidx)


def test_gather_scatter_rows(env):
  torch, de = env
  rng = np.random.default_rng(1)
  for dim, dt in [(64, np.float32), (3, np.float32), (5, np.int8), (1, np.int64), (7, np.float16)]:
    rows = (rng.standard_normal((500, dim)) * 50).astype(dt)
    idx = rng.integers(0, 500, size=2000).astype(np.int32)
    out = de.device_ops.gather_rows(T(torch, rows), T(torch, idx))
    np.testing.assert_array_equal(out.cpu().numpy(), rows[idx])
    perm = rng.permutation(500).astype(np.int32)
    out = de.device_ops.scatter_rows(T(torch, rows), T(torch, perm))
    exp = np.empty_like(rows); exp[perm] = rows
    np.testing.assert_array_equal(out.cpu().numpy(), exp)


@pytest.mark.parametrize("dim", [64, 10, 1])
def test_segment_sum_is_sequential_fp32(env, dim):
  torch, de = env
  rng = np.random.default_rng(dim)
  n = 50000
  ids = (rng.zipf(1.3, size=n) % 4000).astype(np.int64)
  g = rng.standard_normal((n, dim)).astype(np.float32)
  eu, esum, eidx = oopt.segment_sum_by_key(ids, g)
  u, idx, cnt = de.device_ops.unique_no_sync(T(torch, ids))
  out = de.device_ops.segment_sum(T(torch, g), idx, cnt, n)
  U = int(cnt.item())
  assert U == eu.size
  # same summation order as the sequential CPU loop => bit-exact
  np.testing.assert_array_equal(out[:U].cpu().numpy(), esum)


@pytest.mark.parametrize("shards,mode", [(2, 0), (8, 0), (3, 1), (8, 2), (64, 0)])
def test_partition_matches_default_partition_fn(env, shards, mode):
  torch, de = env
  rng = np.random.default_rng(shards)
  n = 70001
  keys = rng.integers(-2**62, 2**62, size=n).astype(np.int64)
  ko, perm, counts = de.device_ops.partition(T(torch, keys), shards, mode)
  if mode == 2:
    # hash mode has no reference twin: check it is a stable partition by SOME owner function
    perm_np = perm.cpu().numpy(); c = counts.cpu().numpy()
    assert c.sum() == n and sorted(perm_np.tolist()) == list(range(n))
    np.testing.assert_array_equal(ko.cpu().numpy(), keys[perm_np])
    off = 0
    for s in range(shards):
      seg = perm_np[off:off + c[s]]
      assert np.all(np.diff(seg) > 0)
      off += c[s]
    return
  owner = ofe.default_partition_fn(keys, shards, gpu_mode=(mode == 0))
  parts, idxs = ofe.make_partition(keys, owner, shards)
  np.testing.assert_array_equal(counts.cpu().numpy(), [len(p) for p in parts])
  np.testing.assert_array_equal(ko.cpu().numpy(), np.concatenate(parts))
  np.testing.assert_array_equal(perm.cpu().numpy(), np.concatenate(idxs))


@pytest.mark.parametrize("n_valid", [0, 1, 255, 256, 40000, 70001, 10**6])
def test_partition_with_device_count(env, n_valid):
  """d_n: only the first min(n, *d_n) keys are partitioned (chains after tfra_unique without a host read)."""
  torch, de = env
  n, shards = 70001, 8
  keys = np.random.default_rng(3).integers(-2**62, 2**62, size=n).astype(np.int64)
  ko, perm, counts = de.device_ops.partition(T(torch, keys), shards, 0, n_dev=torch.tensor(n_valid, device="cuda"))
  m = min(n, n_valid)
  owner = ofe.default_partition_fn(keys[:m], shards, gpu_mode=True)
  parts, idxs = ofe.make_partition(keys[:m], owner, shards)
  np.testing.assert_array_equal(counts.cpu().numpy(), [len(p) for p in parts])
  np.testing.assert_array_equal(ko.cpu().numpy()[:m], np.concatenate(parts) if m else np.zeros(0, np.int64))
  np.testing.assert_array_equal(perm.cpu().numpy()[:m], np.concatenate(idxs) if m else np.zeros(0, np.int32))


@pytest.mark.parametrize("n,dim,hi,zipf", [(1, 4, 5, False), (511, 8, 40, False), (513, 64, 10**9, False),
                                            (131072, 64, 10**8, True), (250000, 16, 2000, True),
                                            (262144, 128, 50, False), (40000, 256, 3, False)])
def test_reduce_by_key(env, n, dim, hi, zipf):
  """unique + unsorted_segment_sum in one call: key set exact, sums vs an fp64 restatement, bit-reproducible per key
  (the ORDER of the distinct keys is unspecified and may differ between calls)."""
  torch, de = env
  rng = np.random.default_rng(n + dim)
  ids = (rng.zipf(1.2, size=n) % hi if zipf else rng.integers(-hi, hi, size=n)).astype(np.int64) * 2654435761
  ids[0] = np.iinfo(np.int64).min
  g = rng.standard_normal((n, dim)).astype(np.float32)
  keys, sums, cnt = de.device_ops.reduce_by_key(T(torch, ids), T(torch, g))
  keys2, sums2, cnt2 = de.device_ops.reduce_by_key(T(torch, ids), T(torch, g))
  u = int(cnt.item())
  ku, inv, counts = np.unique(ids, return_inverse=True, return_counts=True)
  assert u == ku.size == int(cnt2.item())
  k = keys.cpu().numpy()[:u]
  k2 = keys2.cpu().numpy()[:u]
  order, order2 = np.argsort(k), np.argsort(k2)
  assert np.array_equal(k[order], k2[order2])
  assert np.array_equal(sums.cpu().numpy()[:u][order], sums2.cpu().numpy()[:u][order2])   # bit-reproducible sums per key
  np.testing.assert_array_equal(k[order], ku)
  want = np.zeros((ku.size, dim), dtype=np.float64)
  np.add.at(want, inv, g.astype(np.float64))
  got = sums.cpu().numpy()[:u][order]
  tol = 4e-7 * np.sqrt(counts.max()) * 4 + 1e-6
  np.testing.assert_allclose(got, want, rtol=1e-5, atol=tol * np.abs(g).max())
  single = counts == 1                                                  # ids seen once: the row itself
  first = np.zeros(ku.size, dtype=np.int64); first[inv[::-1]] = np.arange(n)[::-1]
  np.testing.assert_array_equal(got[single], g[first[single]])


def test_reduce_by_key_rejects_unsupported(env):
  torch, de = env
  ids = torch.arange(8, device="cuda")
  with pytest.raises(Exception, match="dim % 4"):
    de.device_ops.reduce_by_key(ids, torch.zeros((8, 6), device="cuda"))
  k, s_, c = de.device_ops.reduce_by_key(ids[:0], torch.zeros((0, 8), device="cuda"))
  assert int(c.item()) == 0


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
@pytest.mark.parametrize("dedup", [True, False])
def test_alltoall_route_single_rank(env, backend, dedup):
  """The full id route (unique -> partition -> alltoall ids -> lookup -> alltoall rows -> un-permute ->
  expand; backward mirror) on ONE rank with the collectives forced on: RCCL ("nccl") moves the device
  buffers, "gloo" is the host-staged path.  Result must equal the direct single-table path."""
  torch, de = env
  import torch.distributed as dist
  from tfra_amd.dynamic_embedding.distributed import AllToAllEmbedding
  port = 29900 + (1 if backend == "gloo" else 0) + (2 if dedup else 0)
  dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                          **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
  try:
    rng = np.random.default_rng(11)
    opt = de.optimizers.SGD(0.01)   # linear in the gradient sum: a different (fixed) summation order stays ~1 ulp
    kw = de.DynamicEmbeddingOptimizer.variable_kwargs(opt)
    a = de.Variable(dim=8, name="a2a_a_%s_%d" % (backend, dedup), initializer=0.25, **kw)
    b = de.Variable(dim=8, name="a2a_b_%s_%d" % (backend, dedup), initializer=0.25, **kw)
    oa, ob = de.DynamicEmbeddingOptimizer(opt), de.DynamicEmbeddingOptimizer(de.optimizers.SGD(0.01))
    emb = AllToAllEmbedding(a, partition_mode=0, dedup=dedup, force_collectives=True)
    assert not emb.passthrough
    for step in range(4):
      ids = rng.zipf(1.3, size=(6, 500)).astype(np.int64) % 3000 * 7919
      g = rng.standard_normal((ids.size, 8)).astype(np.float32)
      out = emb.lookup(T(torch, ids))
      ref = b.lookup(T(torch, ids))
      assert out.shape == ref.shape
      np.testing.assert_array_equal(out.cpu().numpy(), ref.cpu().numpy())
      emb.apply_gradients(oa, T(torch, g))
      ob.apply_sparse(b, T(torch, ids), T(torch, g))
    ka, va = a.export(); kb, vb = b.export()
    ia, ib = np.argsort(ka.cpu().numpy()), np.argsort(kb.cpu().numpy())
    np.testing.assert_array_equal(ka.cpu().numpy()[ia], kb.cpu().numpy()[ib])
    # gradient sums are formed in a different (still fixed) order when repeats are summed before routing
    np.testing.assert_allclose(va.cpu().numpy()[ia], vb.cpu().numpy()[ib], rtol=1e-6, atol=1e-6)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_routed_prefetch_step_single_rank(env, backend):
  """RoutedPrefetchStep on ONE rank with the collectives forced on (RCCL moves device buffers, gloo is host-staged): the
  route prepared two batches ahead gives the rows of the direct single-table path bit for bit and the same table after
  training (the gradient sums take the same tree as tfra_reduce_by_key)."""
  torch, de = env
  import torch.distributed as dist
  from tfra_amd.dynamic_embedding.distributed import RoutedPrefetchStep
  port = 29920 + (1 if backend == "gloo" else 0)
  dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                          **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
  try:
    rng = np.random.default_rng(12)
    opt = de.optimizers.Adam(1e-2)
    kw = de.DynamicEmbeddingOptimizer.variable_kwargs(opt)
    a = de.Variable(dim=16, name="rps_a_%s" % backend, initializer=0.25, **kw)
    b = de.Variable(dim=16, name="rps_b_%s" % backend, initializer=0.25, **kw)
    oa, ob = de.DynamicEmbeddingOptimizer(opt), de.DynamicEmbeddingOptimizer(de.optimizers.Adam(1e-2))
    rs = RoutedPrefetchStep(a, oa, partition_mode=0, force_collectives=True)
    assert rs.collectives
    steps = 6
    ids = [T(torch, (rng.zipf(1.25, size=3000).astype(np.int64) % 4000) * 7919 - 5) for _ in range(steps)]
    grads = [T(torch, (rng.standard_normal((3000, 16)) * 0.01).astype(np.float32)) for _ in range(steps)]
    torch.cuda.synchronize()
    rs.feed(ids[0]); rs.feed(ids[1])
    for s in range(steps):
      out = rs.lookup()
      ref = b.lookup(ids[s])
      np.testing.assert_array_equal(out.cpu().numpy(), ref.cpu().numpy())
      rs.apply(grads[s])
      ob.apply_sparse(b, ids[s], grads[s])
      if s + 2 < steps:
        rs.feed(ids[s + 2])
    ka, va = a.export(); kb, vb = b.export()
    ia, ib = np.argsort(ka.cpu().numpy()), np.argsort(kb.cpu().numpy())
    np.testing.assert_array_equal(ka.cpu().numpy()[ia], kb.cpu().numpy()[ib])
    np.testing.assert_allclose(va.cpu().numpy()[ia], vb.cpu().numpy()[ib], rtol=1e-6, atol=1e-6)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("transport,threaded,ahead", [(None, True, 3), ("rccl", True, 3), ("rccl", False, 2), ("staged", True, 1),
                                                      (None, False, 4)])
def test_native_routed_step_single_rank(env, transport, threaded, ahead):
  """The C driver of the routed step (tfra_route_*) on ONE rank: no transport (device copies), its own RCCL
  communicators (grouped ncclSend/ncclRecv to itself) and the host-staged test transport.  Rows bit for bit those of the
  direct single-table path; the same table after training."""
  torch, de = env
  import torch.distributed as dist
  from tfra_amd.dynamic_embedding.distributed import NativeRoutedStep
  backend = {None: None, "rccl": "nccl", "staged": "gloo"}[transport]
  if backend:
    port = 29930 + (1 if backend == "gloo" else 0)
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
  try:
    rng = np.random.default_rng(13)
    opt = de.optimizers.Adam(1e-2)
    kw = de.DynamicEmbeddingOptimizer.variable_kwargs(opt)
    tag = "%s_%d_%d" % (transport, threaded, ahead)
    a = de.Variable(dim=16, name="nrs_a_%s" % tag, initializer=0.25, **kw)
    b = de.Variable(dim=16, name="nrs_b_%s" % tag, initializer=0.25, **kw)
    oa, ob = de.DynamicEmbeddingOptimizer(opt), de.DynamicEmbeddingOptimizer(de.optimizers.Adam(1e-2))
    rs = NativeRoutedStep(a, oa, partition_mode=0, force_collectives=True, max_batch=4096, threaded=threaded)
    steps = 7
    sizes = [3000, 1, 4096, 17, 3000, 2999, 64]
    ids = [T(torch, (rng.zipf(1.25, size=n).astype(np.int64) % 4000) * 7919 - 5) for n in sizes]
    grads = [T(torch, (rng.standard_normal((n, 16)) * 0.01).astype(np.float32)) for n in sizes]
    torch.cuda.synchronize()
    for s in range(ahead):
      rs.feed(ids[s])
    for s in range(steps):
      out = rs.lookup()
      ref = b.lookup(ids[s])
      np.testing.assert_array_equal(out.cpu().numpy(), ref.cpu().numpy())
      rs.apply(grads[s])
      ob.apply_sparse(b, ids[s], grads[s])
      if s + ahead < steps:
        rs.feed(ids[s + ahead])
    with pytest.raises(RuntimeError):
      rs.lookup()
    ka, va = a.export(); kb, vb = b.export()
    ia, ib = np.argsort(ka.cpu().numpy()), np.argsort(kb.cpu().numpy())
    np.testing.assert_array_equal(ka.cpu().numpy()[ia], kb.cpu().numpy()[ib])
    np.testing.assert_allclose(va.cpu().numpy()[ia], vb.cpu().numpy()[ib], rtol=1e-6, atol=1e-6)
    torch.cuda.synchronize()
    rs.close()
  finally:
    if backend:
      dist.destroy_process_group()


def test_k7_k8_sharding(env):
  """K7: default partitioner over 2 shards (T/dynamic_embedding_ops_test.py:324-349);
  K8: custom partitioner keys%2 over 3 shards (:382-408)."""
  torch, de = env
  ids = torch.arange(5, dtype=torch.int64).cuda()
  v = de.Variable(dim=2, devices=["cuda:0", "cuda:0"], name="k7", initializer=0.0)
  emb, tw = de.embedding_lookup(v, ids, return_trainable=True)
  tw.update_op(emb)
  assert int(v.size().item()) == 5 and int(v.size(0).item()) == 3 and int(v.size(1).item()) == 2
  v8 = de.Variable(dim=1, devices=["cuda:0"] * 3, name="k8", partitioner=lambda k, n: (k % 2).to(torch.int32))
  v8.upsert(ids, torch.tensor([[0.], [1.], [2.], [3.], [4.]]).cuda())
  out = v8.lookup(torch.tensor([1, 3, 2, 3, 0], dtype=torch.int64).cuda())
  np.testing.assert_array_equal(out.cpu().numpy(), [[1], [3], [2], [3], [0]])
  assert [int(v8.size(i).item()) for i in range(3)] == [3, 2, 0]


def test_k5_high_rank_and_k9_max_norm(env):
  torch, de = env
  v = de.Variable(dim=1, name="k5", initializer=-1.0)
  v.upsert(torch.tensor([0, 1, 2]).cuda(), torch.tensor([[0.], [1.], [2.]]).cuda())
  out = v.lookup(torch.tensor([[0, 1], [2, 4]]).cuda())
  assert tuple(out.shape) == (2, 2, 1)
  np.testing.assert_array_equal(out.cpu().numpy(), [[[0], [1]], [[2], [-1]]])
  v9 = de.Variable(dim=1, name="k9", initializer=2.0)
  out = de.embedding_lookup(v9, torch.tensor([0]).cuda(), max_norm=1.0)
  np.testing.assert_allclose(out.cpu().numpy(), [[1.0]])
  v92 = de.Variable(dim=2, name="k92", initializer=2.0)
  v92.upsert(torch.tensor([7]).cuda(), torch.tensor([[3.0, 4.0]]).cuda())
  out = de.embedding_lookup(v92, torch.tensor([7]).cuda(), max_norm=2.0)
  np.testing.assert_allclose(out.cpu().numpy(), [[1.2, 1.6]], rtol=1e-6)


def test_k10_initializer_statistics(env):
  """T/dynamic_embedding_variable_test.py:565-588: defaults of 2^17 missing keys follow the
  initializer (random_normal(0, 0.01))."""
  torch, de = env
  gen = torch.Generator(device="cuda").manual_seed(2)
  init = lambda shape: torch.randn(shape, generator=gen, device="cuda") * 0.01
  v = de.Variable(dim=10, name="k10", initializer=init)
  out = v.lookup(torch.arange(2**17, dtype=torch.int64).cuda())
  assert abs(out.mean().item()) < 1e-4 and abs(out.std().item() - 0.01) < 1e-4
  assert int(v.size().item()) == 0  # lookup never inserts


@pytest.mark.parametrize("shards", [1, 3])
def test_embedding_lookup_unique_and_sparse(env, shards):
  torch, de = env
  rng = np.random.default_rng(4)
  dim = 8
  v = de.Variable(dim=dim, devices=["cuda:0"] * shards, name="els%d" % shards, initializer=-2.0)
  cpu = oracle.CpuTable(dim)
  keys = np.arange(0, 300, dtype=np.int64)
  vals = rng.standard_normal((300, dim)).astype(np.float32)
  v.upsert(T(torch, keys), T(torch, vals)); cpu.insert(keys, vals)
  ids = rng.integers(0, 400, size=(7, 31)).astype(np.int64)
  default = np.full(dim, -2.0, np.float32)
  out = de.embedding_lookup_unique(v, T(torch, ids))
  np.testing.assert_array_equal(out.cpu().numpy(), ofe.embedding_lookup_unique(cpu, ids, default))
  rows = np.sort(rng.integers(0, 50, size=600)).astype(np.int64)
  sid = rng.integers(0, 400, size=600).astype(np.int64)
  w = rng.random(600).astype(np.float32) + 0.1
  for comb in ("sum", "mean", "sqrtn"):
    for weights in (None, w):
      got = de.embedding_lookup_sparse(v, (T(torch, rows), T(torch, sid)), None if weights is None else T(torch, weights),
                                       combiner=comb, num_rows=50)
      exp = ofe.embedding_lookup_sparse(cpu, rows, sid, default, comb, weights, num_rows=50)
      np.testing.assert_allclose(got.cpu().numpy(), exp, rtol=2e-6, atol=2e-6)
  with pytest.raises(ValueError, match="combiner"):
    de.embedding_lookup_sparse(v, (T(torch, rows), T(torch, sid)), combiner="max")


OPTS = {
    "sgd": (lambda de: de.optimizers.SGD(0.1), dict(lr=0.1)),
    "adam": (lambda de: de.optimizers.Adam(1e-3, 0.9, 0.999, 1e-8), dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8)),
    "adagrad": (lambda de: de.optimizers.Adagrad(0.05, 0.1), dict(lr=0.05, init_acc=0.1)),
    "adagrad_v2": (lambda de: de.optimizers.Adagrad(0.05, 0.1, 1e-7), dict(lr=0.05, init_acc=0.1, eps=1e-7)),
    "ftrl": (lambda de: de.optimizers.Ftrl(0.05, -0.5, 0.1, 1e-3, 1e-3), dict(lr=0.05, l1=1e-3, l2=1e-3, init_acc=0.1)),
    "ftrl_pow": (lambda de: de.optimizers.Ftrl(0.05, -0.3, 0.1, 0.0, 1e-3), dict(lr=0.05, l1=0.0, l2=1e-3, init_acc=0.1, lr_power=-0.3)),
    # rules without a fused kernel: the reference's own find / dense-apply / upsert sequence (optimizers.Generic)
    "momentum": (lambda de: de.optimizers.Momentum(0.05, 0.9), dict(lr=0.05, momentum=0.9)),
    "momentum_nesterov": (lambda de: de.optimizers.Momentum(0.05, 0.9, use_nesterov=True), dict(lr=0.05, momentum=0.9, nesterov=True)),
    "rmsprop": (lambda de: de.optimizers.RMSProp(0.01, 0.9, 0.5, 1e-10), dict(lr=0.01, rho=0.9, momentum=0.5, eps=1e-10)),
}


@pytest.mark.parametrize("name", list(OPTS))
@pytest.mark.parametrize("dim,shards", [(1, 1), (10, 2), (64, 1)])
def test_k13_fused_optimizer_matches_reference_sequence(env, name, dim, shards):
  """K13 (T/dynamic_embedding_optimizer_test.py:546-641): ids [0,1,1,2,3,4,4] with duplicates
  summed, 10 steps, shards {1,2}, dim {1,10} — fused HIP update vs the reference's
  (1+S) finds + dense apply + (1+S) upserts over CPU tables.  Tolerance 1e-6 (north_star)."""
  torch, de = env
  kind = name.split("_")[0]
  mk, hyper = OPTS[name]
  opt = mk(de)
  deo = de.DynamicEmbeddingOptimizer(opt)
  v = de.Variable(dim=dim, devices=["cuda:0"] * shards, name="k13_%s_%d_%d" % (name, dim, shards), initializer=0.25,
                  **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  rng = np.random.default_rng(13)
  base_keys = np.arange(5, dtype=np.int64)
  base = rng.random((5, dim)).astype(np.float32)
  v.upsert(T(torch, base_keys), T(torch, base))
  nslots = len(opt.slots)
  tabs = [oracle.CpuTable(dim) for _ in range(1 + nslots)]
  tabs[0].insert(base_keys, base)
  ora = oopt.SparseOptimizerOracle(kind, tabs[0], tabs[1:], hyper, 0.25)
  ids = np.array([0, 1, 1, 2, 3, 4, 4, 9], dtype=np.int64)  # 9 is unseen: inserted by the write-back
  for step in range(10):
    emb, tw = de.embedding_lookup(v, T(torch, ids), return_trainable=True)
    exp = tabs[0].find(ids, np.full(dim, 0.25, np.float32))
    np.testing.assert_allclose(emb.cpu().numpy(), exp, rtol=1e-6, atol=1e-6)
    g = (2.0 * exp * rng.random((ids.size, 1)).astype(np.float32)).astype(np.float32)  # d/dE (E x)^2-like
    deo.apply_gradients([(T(torch, g), tw)])
    ora.apply(ids, g)
  k, val = v.export()
  o = np.argsort(k.cpu().numpy())
  ek, ev = tabs[0].export_sorted()
  np.testing.assert_array_equal(k.cpu().numpy()[o], ek)
  # general lr_power goes through powf, which neither libm nor the device library rounds
  # correctly: 5e-6 there, 1e-6 (north_star) everywhere else
  tol = 5e-6 if name == "ftrl_pow" else 1e-6
  np.testing.assert_allclose(val.cpu().numpy()[o], ev, rtol=tol, atol=tol)
  for si, sname in enumerate(opt.slots):
    got = deo.get_slot(v, sname).lookup(T(torch, ek)).cpu().numpy()
    exp = tabs[1 + si].find(ek, np.zeros(dim, np.float32))
    np.testing.assert_allclose(got, exp, rtol=tol, atol=tol)


@pytest.mark.parametrize("kat_i", range(4))
@pytest.mark.parametrize("path", ["apply_gradients", "apply_optimizer"])
def test_ftrl_tensorflow_known_answers(env, kat_i, path):
  """The four FTRL known answers TensorFlow's own ftrl_test.py publishes (tests/test_optimizers_pinned.py::FTRL_KATS) through
  the HIP kernels: fused planned write-back (apply_gradients) and tfra_table_apply_optimizer on pre-summed unique keys.
  Two resident keys = TF's var0 / var1 (dim 2), constant gradients, lr 3.0, initial_accumulator_value 0.1."""
  torch, de = env
  from tests.test_optimizers_pinned import FTRL_KATS, FTRL_KAT_GRADS
  name, v0, v1, l1, l2, steps, e0, e1 = FTRL_KATS[kat_i]
  opt = de.optimizers.Ftrl(3.0, -0.5, 0.1, l1, l2)
  deo = de.DynamicEmbeddingOptimizer(opt)
  v = de.Variable(dim=2, name="ftrl_kat_%d_%s" % (kat_i, path), initializer=0.0, **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  keys = torch.tensor([7, 11], dtype=torch.int64, device="cuda")
  v.upsert(keys, torch.tensor([v0, v1], dtype=torch.float32, device="cuda"))
  g = torch.tensor(FTRL_KAT_GRADS, dtype=torch.float32, device="cuda")
  for _ in range(steps):
    if path == "apply_gradients":
      _, tw = de.embedding_lookup(v, keys, return_trainable=True)
      deo.apply_gradients([(g, tw)])
    else:
      v.tables[0]._table.apply_optimizer(deo.begin_step(), keys, g, torch.zeros(2, device="cuda"))
  got = v.lookup(keys).cpu().numpy()
  np.testing.assert_allclose(got, np.array([e0, e1], np.float32), rtol=1e-5, atol=0)


def test_generic_optimizer_custom_rule_and_errors(env):
  """optimizers.Generic with a user rule (here Adamax, T/dynamic_embedding_optimizer_test.py:313-319 lists it):
  same write-back sequence; a plan cannot be built for it; slot count is limited by the row layout."""
  torch, de = env

  def adamax(step, p, g, m, v, lr=0.01, b1=0.9, b2=0.999, eps=1e-7):
    m = m * b1 + g * (1 - b1)
    v = torch.maximum(v * b2, g.abs())
    return p - (lr / (1 - b1 ** step)) * m / (v + eps), m, v

  opt = de.optimizers.Generic(slots=("m", "v"), slot_init=(0.0, 0.0), update=adamax)
  deo = de.DynamicEmbeddingOptimizer(opt)
  var = de.Variable(dim=8, name="generic_adamax", initializer=0.5, **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  rng = np.random.default_rng(2)
  ref = {}
  for step in range(1, 6):
    ids = rng.integers(0, 30, size=64).astype(np.int64)
    g = rng.standard_normal((64, 8)).astype(np.float32)
    deo.apply_sparse(var, T(torch, ids), T(torch, g))
    uniq, inv = np.unique(ids, return_inverse=True)
    gs = np.zeros((uniq.size, 8), np.float64); np.add.at(gs, inv, g.astype(np.float64))
    for k, gk in zip(uniq.tolist(), gs.astype(np.float32)):
      p, m, v = ref.get(k, (np.full(8, 0.5, np.float32), np.zeros(8, np.float32), np.zeros(8, np.float32)))
      m = m * 0.9 + gk * 0.1
      v = np.maximum(v * 0.999, np.abs(gk))
      ref[k] = ((p - (0.01 / (1 - 0.9 ** step)) * m / (v + 1e-7)).astype(np.float32), m.astype(np.float32), v.astype(np.float32))
  ks = np.array(sorted(ref), dtype=np.int64)
  got = var.lookup(T(torch, ks)).cpu().numpy()
  np.testing.assert_allclose(got, np.stack([ref[int(k)][0] for k in ks]), rtol=2e-5, atol=2e-6)
  np.testing.assert_allclose(deo.get_slot(var, "v").lookup(T(torch, ks)).cpu().numpy(),
                             np.stack([ref[int(k)][2] for k in ks]), rtol=2e-5, atol=2e-6)
  with pytest.raises(ValueError):
    deo.plan(var, T(torch, ks))
  with pytest.raises(ValueError):
    de.optimizers.Generic(slots=("a", "b", "c", "d", "e"))


def test_optimizer_requires_slots(env):
  torch, de = env
  v = de.Variable(dim=4, name="noslots")
  deo = de.DynamicEmbeddingOptimizer(de.optimizers.Adam())
  with pytest.raises(ValueError, match="aux_fields"):
    deo.apply_sparse(v, torch.tensor([1]).cuda(), torch.zeros(1, 4).cuda())


def test_bp_v2_accum_write_back(env):
  """bp_v2: update_op sends where(exists, new-old, new) through accum (PY/embedding_weights.py:434-444)."""
  torch, de = env
  v = de.Variable(dim=3, name="bpv2", initializer=1.0, bp_v2=True)
  v.upsert(torch.tensor([1, 2]).cuda(), torch.tensor([[1., 1, 1], [2, 2, 2]]).cuda())
  emb, tw = de.embedding_lookup(v, torch.tensor([1, 2, 3]).cuda(), return_trainable=True)
  np.testing.assert_array_equal(tw.exists.cpu().numpy(), [True, True, False])
  v.upsert(torch.tensor([1]).cuda(), torch.tensor([[10., 10, 10]]).cuda())  # someone else moved key 1
  tw.update_op(emb + 0.5)
  out = v.lookup(torch.tensor([1, 2, 3]).cuda()).cpu().numpy()
  np.testing.assert_array_equal(out, [[10.5] * 3, [2.5] * 3, [1.5] * 3])


@pytest.mark.parametrize("dim", [64, 8, 128, 200])
@pytest.mark.parametrize("kind", ["adam", "ftrl", "adagrad", "sgd"])
def test_apply_sparse_two_kernel_path_zipf(env, dim, kind):
  """The fused tile-reduce + bucket-apply path on Zipf batches with heavy duplication (hot key
  ~18 % of the batch) vs the reference sequence over CPU tables.  Sums of duplicates use a fixed
  tree instead of the sequential order => compare at 1e-6 relative to the summed-gradient scale
  (fp64 oracle sums), and require bit-identical results across two runs (determinism)."""
  torch, de = env
  from bench import zipf_bounded, keys_of_ranks
  mk, hyper = OPTS[kind]
  rng = np.random.default_rng(dim)
  n_keys, B = 50000, 20000 + dim  # not a multiple of the 1024 tile
  results = []
  for run in range(2):
    opt = mk(de)
    deo = de.DynamicEmbeddingOptimizer(opt)
    v = de.Variable(dim=dim, name="as_%s_%d_%d" % (kind, dim, run), initializer=0.1,
                    **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    tabs = [oracle.CpuTable(dim) for _ in range(1 + len(opt.slots))]
    ora = oopt.SparseOptimizerOracle(kind, tabs[0], tabs[1:], hyper, 0.1)
    r2 = np.random.default_rng(1000 + dim)
    for step in range(3):
      ids = keys_of_ranks(zipf_bounded(r2, B, n_keys))
      g = (r2.standard_normal((B, dim)) * 0.01).astype(np.float32)
      deo.apply_sparse(v, T(torch, ids), T(torch, g))
      if run == 0:
        # oracle with fp64 duplicate sums (order-free), rounded once to fp32
        uniq, inv = np.unique(ids, return_inverse=True)
        gs = np.zeros((uniq.size, dim), np.float64); np.add.at(gs, inv, g.astype(np.float64))
        ora.apply(uniq, gs.astype(np.float32))
    k, val = v.export()
    o = np.argsort(k.cpu().numpy())
    results.append((k.cpu().numpy()[o], val.cpu().numpy()[o]))
    if run == 0:
      ek, ev = tabs[0].export_sorted()
      np.testing.assert_array_equal(results[0][0], ek)
      # Adam/FTRL normalise the update: errors of the gradient sum shrink further; 2e-6 abs covers
      # the hot key whose 3600-term sum differs from the fp64 sum by ~1e-6 relative
      np.testing.assert_allclose(results[0][1], ev, rtol=2e-6, atol=2e-6)
      assert int(v.size().item()) == ek.size
  np.testing.assert_array_equal(results[0][0], results[1][0])
  np.testing.assert_array_equal(results[0][1], results[1][1])


def test_apply_sparse_matches_exact_order_path_without_duplicates(env):
  """With unique ids there is nothing to sum: fused path == exact path bit for bit."""
  torch, de = env
  rng = np.random.default_rng(8)
  dim = 64
  ids = rng.permutation(10**6)[:30000].astype(np.int64)
  g = (rng.standard_normal((ids.size, dim)) * 0.01).astype(np.float32)
  outs = []
  for exact in (False, True):
    opt = de.optimizers.Adam(1e-3)
    deo = de.DynamicEmbeddingOptimizer(opt, exact_order=exact)
    v = de.Variable(dim=dim, name="ex%d" % exact, initializer=0.3, **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    for _ in range(2):
      deo.apply_sparse(v, T(torch, ids), T(torch, g))
    k, val = v.export()
    o = np.argsort(k.cpu().numpy())
    outs.append(val.cpu().numpy()[o])
  np.testing.assert_array_equal(outs[0], outs[1])


def test_apply_sparse_hot_bucket_multipass(env):
  """100 hot keys that all hash into ONE merge bucket and occur in every tile: the bucket holds
  more descriptors than fit in LDS and is merged in several hash-split passes."""
  torch, de = env
  from bench import fmix64_np
  rng = np.random.default_rng(77)
  n, dim = 40064, 8
  P = 64
  while P < 2048 and P * 128 < n:
    P *= 2
  cand = rng.integers(1, 2**62, size=200000).astype(np.int64)
  h = fmix64_np(cand.astype(np.uint64))
  bucket = ((h >> np.uint64(32)).astype(np.uint64) * np.uint64(P)) >> np.uint64(32)  # ~ mulhi(h, P)
  hot = cand[bucket == 7][:100]
  assert hot.size == 100
  ids = np.concatenate([np.tile(hot, n // 100), hot[: n % 100]])
  rng.shuffle(ids)
  g = (rng.standard_normal((n, dim)) * 0.01).astype(np.float32)
  opt = de.optimizers.Adagrad(0.05, 0.1)
  deo = de.DynamicEmbeddingOptimizer(opt)
  v = de.Variable(dim=dim, name="hotbucket", initializer=0.2, **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  deo.apply_sparse(v, T(torch, ids), T(torch, g))
  tabs = [oracle.CpuTable(dim), oracle.CpuTable(dim)]
  ora = oopt.SparseOptimizerOracle("adagrad", tabs[0], tabs[1:], dict(lr=0.05, init_acc=0.1), 0.2)
  uniq, inv = np.unique(ids, return_inverse=True)
  gs = np.zeros((uniq.size, dim), np.float64); np.add.at(gs, inv, g.astype(np.float64))
  ora.apply(uniq, gs.astype(np.float32))
  assert int(v.size().item()) == 100
  k, val = v.export()
  o = np.argsort(k.cpu().numpy())
  ek, ev = tabs[0].export_sorted()
  np.testing.assert_array_equal(k.cpu().numpy()[o], ek)
  np.testing.assert_allclose(val.cpu().numpy()[o], ev, rtol=2e-6, atol=2e-6)


def _hot_bucket_ids(rng, n):
  from bench import fmix64_np
  P = 64
  while P < 2048 and P * 128 < n:
    P *= 2
  cand = rng.integers(1, 2**62, size=200000).astype(np.int64)
  h = fmix64_np(cand.astype(np.uint64))
  bucket = ((h >> np.uint64(32)).astype(np.uint64) * np.uint64(P)) >> np.uint64(32)
  hot = cand[bucket == 7][:100]
  ids = np.concatenate([np.tile(hot, n // 100), hot[: n % 100]])
  rng.shuffle(ids)
  return ids


@pytest.mark.parametrize("kind,dim,n,shape", [("adam", 64, 131072, "zipf"), ("ftrl", 8, 3001, "zipf"),
                                               ("adagrad", 128, 20000, "uniform"), ("sgd", 256, 5000, "zipf"),
                                               ("adagrad", 8, 40064, "hot_bucket"), ("adam", 16, 1, "uniform"),
                                               ("adam", 32, 262144, "zipf")])
def test_planned_write_back_is_bit_identical(env, kind, dim, n, shape):
  """tfra_sparse_plan_build (on a side stream) + tfra_table_apply_planned == tfra_table_apply_sparse, bit
  for bit, over several steps with the plan object reused — same keys, same rows, same slots."""
  torch, de = env
  from bench import zipf_bounded, keys_of_ranks
  rng = np.random.default_rng(n + dim)
  mk = {"sgd": lambda: de.optimizers.SGD(0.1), "adam": lambda: de.optimizers.Adam(0.01),
        "adagrad": lambda: de.optimizers.Adagrad(0.05, 0.1),
        "ftrl": lambda: de.optimizers.Ftrl(0.05, l1_regularization_strength=1e-3, l2_regularization_strength=1e-3)}[kind]
  opt_a, opt_b = mk(), mk()
  kw = de.DynamicEmbeddingOptimizer.variable_kwargs(opt_a)
  name = "plan_%s_%d_%d_%s" % (kind, dim, n, shape)
  a = de.Variable(dim=dim, name=name + "_a", initializer=0.3, **kw)
  b = de.Variable(dim=dim, name=name + "_b", initializer=0.3, **kw)
  da, db = de.DynamicEmbeddingOptimizer(opt_a), de.DynamicEmbeddingOptimizer(opt_b)
  side = torch.cuda.Stream()
  plan = None
  for step in range(3):
    if shape == "zipf":
      ids = keys_of_ranks(zipf_bounded(rng, n, 10**7))
    elif shape == "hot_bucket":
      ids = _hot_bucket_ids(rng, n)
    else:
      ids = rng.integers(-10**6, 10**6, size=n).astype(np.int64)
    g = (rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
    ids_t, g_t = T(torch, ids), T(torch, g)
    da.apply_sparse(a, ids_t, g_t)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      plan = db.plan(b, ids_t, plan)
    out = b.lookup(ids_t[: min(n, 1000)])            # the lookup of the step runs next to the plan build
    db.apply_sparse(b, ids_t, g_t, plan=plan)
    assert out.shape[-1] == dim
  assert int(a.size()) == int(b.size())
  ka, va = a.export()
  kb, vb = b.export()
  ia, ib = np.argsort(ka.cpu().numpy()), np.argsort(kb.cpu().numpy())
  np.testing.assert_array_equal(ka.cpu().numpy()[ia], kb.cpu().numpy()[ib])
  np.testing.assert_array_equal(va.cpu().numpy()[ia], vb.cpu().numpy()[ib])
  for slot in opt_a.slots:                            # optimizer state vectors too
    sa = da.get_slot(a, slot).lookup(ka[ia])
    sb = db.get_slot(b, slot).lookup(ka[ia])
    assert torch.equal(sa, sb)


def test_plan_rejects_mismatches(env):
  torch, de = env
  opt = de.optimizers.SGD(0.1)
  deo = de.DynamicEmbeddingOptimizer(opt)
  v8 = de.Variable(dim=8, name="plan_rej8", initializer=0.0)
  v6 = de.Variable(dim=6, name="plan_rej6", initializer=0.0)
  v16 = de.Variable(dim=16, name="plan_rej16", initializer=0.0)
  ids = torch.arange(10, device="cuda")
  with pytest.raises(ValueError):
    deo.plan(v6, ids)                                 # dim % 4 != 0 takes the unique + segment_sum path
  plan = deo.plan(v8, ids)
  with pytest.raises(ValueError):
    deo.apply_sparse(v16, ids, torch.zeros((10, 16), device="cuda"), plan=plan)
  with pytest.raises(ValueError):
    deo.apply_sparse(v8, ids, torch.zeros((9, 8), device="cuda"), plan=plan)
  empty = deo.plan(v8, ids[:0])
  deo.apply_sparse(v8, ids[:0], torch.zeros((0, 8), device="cuda"), plan=empty)
  assert int(v8.size()) == 0


def test_captured_train_step_matches_eager(env):
  """HIP-graph replay of lookup + sparse Adam == the same steps launched eagerly (bit for bit),
  including Adam's per-step lr_t fed through device memory."""
  torch, de = env
  from bench import zipf_bounded, keys_of_ranks
  dim, B, n_keys = 64, 8192, 30000
  rng = np.random.default_rng(21)
  batches = [keys_of_ranks(zipf_bounded(rng, B, n_keys)) for _ in range(5)]
  g = (rng.standard_normal((B, dim)) * 0.01).astype(np.float32)
  outs = []
  for mode in ("eager", "graph"):
    opt = de.optimizers.Adam(1e-2)
    deo = de.DynamicEmbeddingOptimizer(opt)
    v = de.Variable(dim=dim, name="cap_" + mode, initializer=0.05, init_size=200000,
                    **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    looks = []
    if mode == "eager":
      for _ in range(2):  # the captured variant runs 2 eager warm-up steps on batch 0 ...
        deo.apply_sparse(v, T(torch, batches[0]), T(torch, g))
      deo.iterations += 1  # ... and spends one step number on the capture itself (not executed)
      for b in batches:
        looks.append(v.lookup(T(torch, b)).cpu().numpy())
        deo.apply_sparse(v, T(torch, b), T(torch, g))
    else:
      cap = de.CapturedTrainStep(v, deo, B)
      cap.grads.copy_(T(torch, g))
      cap.capture(warmup_ids=T(torch, batches[0]))
      for b in batches:
        looks.append(cap.step(T(torch, b)).cpu().numpy().copy())
      cap.close()
    k, val = v.export()
    o = np.argsort(k.cpu().numpy())
    outs.append((k.cpu().numpy()[o], val.cpu().numpy()[o], looks))
  np.testing.assert_array_equal(outs[0][0], outs[1][0])
  np.testing.assert_array_equal(outs[0][1], outs[1][1])
  for a, b in zip(outs[0][2], outs[1][2]):
    np.testing.assert_array_equal(a, b)


def test_prefetch_step_driver_matches_eager(env):
  """tfra_table_step_prefetch (lookup + gradient half on the main stream, plan of the next batch on a second
  stream, one C call per step) == eager lookup + apply_sparse, bit for bit."""
  torch, de = env
  from bench import zipf_bounded, keys_of_ranks
  dim, B, n_keys = 64, 20000, 50000
  rng = np.random.default_rng(23)
  batches = [keys_of_ranks(zipf_bounded(rng, B, n_keys)) for _ in range(7)]
  grads = [(rng.standard_normal((B, dim)) * 0.01).astype(np.float32) for _ in range(7)]
  outs = []
  for mode in ("eager", "driver"):
    opt = de.optimizers.Adam(1e-2)
    deo = de.DynamicEmbeddingOptimizer(opt)
    v = de.Variable(dim=dim, name="pfd_" + mode, initializer=0.05, **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    looks = []
    if mode == "eager":
      for b, g in zip(batches, grads):
        looks.append(v.lookup(T(torch, b)).cpu().numpy())
        deo.apply_sparse(v, T(torch, b), T(torch, g))
    else:
      ps = de.PrefetchStep(v, deo).prime(T(torch, batches[0]))
      for i, g in enumerate(grads):
        nxt = T(torch, batches[i + 1]) if i + 1 < len(batches) else None
        looks.append(ps.step(T(torch, g), nxt).cpu().numpy())
    assert deo.iterations == len(batches)
    k, val = v.export()
    o = np.argsort(k.cpu().numpy())
    outs.append((k.cpu().numpy()[o], val.cpu().numpy()[o], looks))
  np.testing.assert_array_equal(outs[0][0], outs[1][0])
  np.testing.assert_array_equal(outs[0][1], outs[1][1])
  for a, b in zip(outs[0][2], outs[1][2]):
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("nplans", [2, 3, 4])
def test_prefetch_step_driver_ragged_and_few_plans(env, nplans):
  """Batch sizes change from step to step (incl. an empty batch), only `nplans` plans rotate (2 forces the
  host-side fallback waits of the driver), other table ops run between the steps: still bit-identical."""
  torch, de = env
  from bench import zipf_bounded, keys_of_ranks
  dim, n_keys = 32, 20000
  rng = np.random.default_rng(100 + nplans)
  sizes = [5000, 1, 70000, 0, 513, 131072, 12, 30000, 30000, 7]
  batches = [keys_of_ranks(zipf_bounded(rng, max(n, 1), n_keys))[:n] for n in sizes]
  grads = [(rng.standard_normal((n, dim)) * 0.01).astype(np.float32) for n in sizes]
  outs = []
  for mode in ("eager", "driver"):
    opt = de.optimizers.Adagrad(0.05, 0.1)
    deo = de.DynamicEmbeddingOptimizer(opt)
    v = de.Variable(dim=dim, name="pfr_%s_%d" % (mode, nplans), initializer=0.05,
                    **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    looks = []
    if mode == "eager":
      for b, g in zip(batches, grads):
        looks.append(v.lookup(T(torch, b)).cpu().numpy())
        if b.size:
          deo.apply_sparse(v, T(torch, b), T(torch, g))
        else:
          deo.iterations += 1
        looks.append(v.lookup(T(torch, batches[0][:100])).cpu().numpy())   # an unrelated op between the steps
    else:
      ps = de.PrefetchStep(v, deo)
      ps.NPLANS = nplans
      ps.plans = ps.plans[:nplans] if nplans <= len(ps.plans) else ps.plans
      ps.ids = [None] * nplans
      ps.prime(T(torch, batches[0]))
      for i, g in enumerate(grads):
        nxt = T(torch, batches[i + 1]) if i + 1 < len(batches) else None
        looks.append(ps.step(T(torch, g), nxt).cpu().numpy())
        looks.append(v.lookup(T(torch, batches[0][:100])).cpu().numpy())
    assert deo.iterations == len(batches)
    k, val = v.export()
    o = np.argsort(k.cpu().numpy())
    outs.append((k.cpu().numpy()[o], val.cpu().numpy()[o], looks))
  np.testing.assert_array_equal(outs[0][0], outs[1][0])
  np.testing.assert_array_equal(outs[0][1], outs[1][1])
  for a, b in zip(outs[0][2], outs[1][2]):
    np.testing.assert_array_equal(a, b)


def test_prefetch_step_driver_with_growth_and_bounded_table(env):
  """The step driver on (a) a table that rehashes several times while steps are in flight and (b) a bounded
  Hkv table at capacity (the gradient half evicts): same results as the plain calls / capacity respected."""
  torch, de = env
  dim = 16
  rng = np.random.default_rng(31)
  # (a) growth: 12 steps of 40 000 mostly-new keys into a table that starts with 64 slots
  batches = [np.concatenate([rng.integers(0, 10**9, size=39000), rng.integers(0, 500, size=1000)]).astype(np.int64) for _ in range(12)]
  grads = [(rng.standard_normal((40000, dim)) * 0.1).astype(np.float32) for _ in range(12)]
  outs = []
  for mode in ("eager", "driver"):
    opt = de.optimizers.Adam(1e-2)
    deo = de.DynamicEmbeddingOptimizer(opt)
    v = de.Variable(dim=dim, name="pfg_" + mode, initializer=0.1, init_size=64, **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    if mode == "eager":
      for b, g in zip(batches, grads):
        v.lookup(T(torch, b))
        deo.apply_sparse(v, T(torch, b), T(torch, g))
    else:
      ps = de.PrefetchStep(v, deo).prime(T(torch, batches[0]))
      for i, g in enumerate(grads):
        ps.step(T(torch, g), T(torch, batches[i + 1]) if i + 1 < len(batches) else None)
    k, val = v.export()
    o = np.argsort(k.cpu().numpy())
    outs.append((k.cpu().numpy()[o], val.cpu().numpy()[o]))
  assert outs[0][0].size > 400000
  np.testing.assert_array_equal(outs[0][0], outs[1][0])
  np.testing.assert_array_equal(outs[0][1], outs[1][1])
  # (b) bounded table, LRU: every step's keys are resident afterwards, size never exceeds the capacity
  opt = de.optimizers.SGD(0.5)
  deo = de.DynamicEmbeddingOptimizer(opt)
  v = de.get_variable("pfb_hkv", key_dtype=torch.int64, value_dtype=torch.float32, initializer=1.0, dim=dim, init_size=4096,
                      kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(
                          init_capacity=4096, max_capacity=4096, max_hbm_for_values=1 << 22,
                          evict_strategy=de.HkvEvictStrategy.LRU)))
  steps = [np.arange(i * 1500, i * 1500 + 1500, dtype=np.int64) * 7 + 1 for i in range(8)]
  ps = de.PrefetchStep(v, deo).prime(T(torch, steps[0]))
  for i in range(len(steps)):
    ps.step(torch.ones((1500, dim), device="cuda"), T(torch, steps[i + 1]) if i + 1 < len(steps) else None)
    assert int(v.size()) <= 4096
    got, ex = v.lookup(T(torch, steps[i]), return_exists=True)
    assert bool(ex.all())
    np.testing.assert_array_equal(got.cpu().numpy(), np.full((1500, dim), 0.5, np.float32))


def test_dynamic_partition_reference_kats(env):
  """The reference's own DynamicPartition cases (T/dynamic_partition_op_test.py) on the device ops, incl. the
  GPU kernel's discard-out-of-range behaviour."""
  torch, de = env
  from tests import kats_partition
  for name, data, parts, num, want in kats_partition.partition_cases():
    got = de.device_ops.dynamic_partition(T(torch, data), T(torch, parts), num)
    assert len(got) == num, name
    for g, w in zip(got, want):
      g = g.cpu().numpy()
      w = np.asarray(w, dtype=np.float32).reshape(g.shape) if np.asarray(w).size == 0 else np.asarray(w)
      np.testing.assert_array_equal(g, w, err_msg=name)
  with pytest.raises(ValueError):      # testErrorWrongDimsIndices :324-329
    de.device_ops.dynamic_partition(T(torch, np.zeros((3, 1), np.float32)), T(torch, np.zeros((2, 1), np.int32)), 4)


def test_dynamic_stitch_reference_kats(env):
  """The reference's own DynamicStitch cases (T/dynamic_stitch_op_test.py) + partition/stitch round trip."""
  torch, de = env
  from tests import kats_partition
  for name, idx, data, want in kats_partition.stitch_cases():
    got = de.device_ops.dynamic_stitch([torch.from_numpy(np.asarray(i)) for i in idx],
                                       [T(torch, np.asarray(d)) for d in data]).cpu().numpy()
    np.testing.assert_array_equal(got, np.asarray(want).reshape(got.shape), err_msg=name)
  with pytest.raises(ValueError):      # testErrorDataDimSizeMismatch :197-208
    de.device_ops.dynamic_stitch([torch.tensor([0, 4, 5]), torch.tensor([1, 6, 2, 3])],
                                 [T(torch, np.zeros((3, 2), np.float32)), T(torch, np.zeros((4, 3), np.float32))])
  # dynamic_stitch(dynamic_partition(range(n)), dynamic_partition(data)) == data  (PY/..._variable.py:131-162)
  rng = np.random.default_rng(4)
  n, shards = 50001, 7
  data = rng.standard_normal((n, 5)).astype(np.float32)
  owner = rng.integers(0, shards, size=n).astype(np.int32)
  parts = de.device_ops.dynamic_partition(T(torch, data), T(torch, owner), shards)
  idxs = de.device_ops.dynamic_partition(torch.arange(n, dtype=torch.int32, device="cuda"), T(torch, owner), shards)
  back = de.device_ops.dynamic_stitch(idxs, parts)
  np.testing.assert_array_equal(back.cpu().numpy(), data)


@pytest.mark.parametrize("shards", [1, 3])
def test_safe_embedding_lookup_sparse_reference_kats(env, shards):
  """T/dynamic_embedding_ops_test.py:1007-1165 (2-D, incl. `partitioned`) and :1205-1324 (3-D): ids are never
  pruned (-100 is a key), weights <= 0 are, empty rows -> zeros / `default_id`."""
  torch, de = env
  from tests import kats_sparse
  E = kats_sparse.embeddings(np.random.default_rng(shards))
  v = de.Variable(dim=kats_sparse.DIM, devices=["cuda:0"] * shards, name="safe_kat_%d" % shards, initializer=0.0)
  ks = np.array(sorted(E), dtype=np.int64)
  v.upsert(T(torch, ks), T(torch, np.stack([E[int(k)] for k in ks])))

  def lookup(idx, ids, shape, w, default_id):
    return de.safe_embedding_lookup_sparse(v, (np.asarray(idx), np.asarray(ids, np.int64), shape),
                                           None if w is None else np.asarray(w, np.float32),
                                           default_id=default_id).cpu().numpy()

  kats_sparse.run(lookup, E)
  # combiner "sum" keeps non-positive weights (`_prune_invalid_weights` is skipped, :374-376)
  got = de.safe_embedding_lookup_sparse(v, (np.asarray(kats_sparse.IDX_2D), np.asarray(kats_sparse.IDS, np.int64),
                                            kats_sparse.SHAPE_2D), np.asarray(kats_sparse.WEIGHTS, np.float32),
                                        combiner="sum").cpu().numpy()
  np.testing.assert_allclose(got[4], 0.0 * E[0] - 0.5 * E[1], rtol=1e-6, atol=1e-6)
  np.testing.assert_allclose(got[0], E[0] + 2 * E[1] + E[-100], rtol=1e-6, atol=1e-6)


def test_sparse_segment_sum_reference_kats(env):
  """T/math_ops_test.py:60-131 (de.math.sparse_segment_sum with / without num_segments)."""
  torch, de = env
  data = T(torch, np.array([[1, 2, 3, 4], [-1, -2, -3, -4], [5, 6, 7, 8]], np.float32))
  idx = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
  seg = torch.tensor([0, 5], dtype=torch.int64, device="cuda")
  for n in (6, 100):
    got = de.device_ops.sparse_segment_combine(data, idx, seg, None, "sum", n).cpu().numpy()
    want = np.zeros((n, 4), np.float32)
    want[0], want[5] = [1, 2, 3, 4], [-1, -2, -3, -4]
    np.testing.assert_array_equal(got, want)
  with pytest.raises(ValueError):      # 6 indices vs 7 segment ids
    de.device_ops.sparse_segment_combine(T(torch, np.arange(20, dtype=np.float32).reshape(10, 2)),
                                         torch.arange(6, dtype=torch.int32, device="cuda"),
                                         torch.arange(7, dtype=torch.int64, device="cuda"), None, "sum", 100)


def test_write_back_plan_started_at_lookup_time(env):
  """embedding_lookup(..., return_trainable=True, plan_writeback=True) + apply_gradients — the reference's API sequence —
  builds the id-only half of the write-back on a second stream from lookup time on (the TrainableWrapper holds the ids):
  same table, bit for bit, as lookup + one-call apply_sparse; plans go back to the Variable's pool, also when a wrapper
  is dropped unapplied."""
  torch, de = env
  rng = np.random.default_rng(21)
  opt = de.optimizers.Adam(1e-2)
  kw = de.DynamicEmbeddingOptimizer.variable_kwargs(opt)
  a = de.Variable(dim=32, name="plan_at_lookup_a", initializer=0.1, **kw)
  b = de.Variable(dim=32, name="plan_at_lookup_b", initializer=0.1, **kw)
  oa, ob = de.DynamicEmbeddingOptimizer(opt), de.DynamicEmbeddingOptimizer(de.optimizers.Adam(1e-2))
  n = 20_000
  for step in range(4):
    ids = T(torch, (rng.zipf(1.2, size=n).astype(np.int64) % 30_000) * 7919 - 3).reshape(100, 200)
    g = T(torch, (rng.standard_normal((n, 32)) * 0.01).astype(np.float32))
    emb, tw = de.embedding_lookup(a, ids, return_trainable=True, plan_writeback=True)
    assert tw.plan is not None and emb.shape == (100, 200, 32)
    ref = b.lookup(ids.reshape(-1))
    np.testing.assert_array_equal(emb.reshape(n, 32).cpu().numpy(), ref.cpu().numpy())
    if step == 2:   # a lookup whose gradients never come: its plan returns to the pool with the wrapper
      _, dropped = de.embedding_lookup(a, ids, return_trainable=True, plan_writeback=True)
      assert dropped.plan is not None
      del dropped
    oa.apply_gradients([(g, tw)])
    assert tw.plan is None
    ob.apply_sparse(b, ids.reshape(-1), g)
  pool = a._plan_pool
  assert pool["made"] == 2 and len(pool["free"]) == 2
  small, tws = de.embedding_lookup(a, ids.reshape(-1)[:100], return_trainable=True, plan_writeback=True)
  assert tws.plan is None                      # below PLAN_AT_LOOKUP_MIN_IDS
  _, plain_tw = de.embedding_lookup(a, ids, return_trainable=True)
  assert plain_tw.plan is None and len(pool["free"]) == 2   # not asked for: nothing planned
  ka, va = a.export(); kb, vb = b.export()
  ia, ib = torch.argsort(ka), torch.argsort(kb)
  assert torch.equal(ka[ia], kb[ib]) and torch.equal(va[ia], vb[ib])
  for slot in ("m", "v"):
    assert torch.equal(oa.get_slot(a, slot).lookup(ka[ia]), ob.get_slot(b, slot).lookup(kb[ib]))


@pytest.mark.parametrize("vdtype", ["float16", "bfloat16"])
@pytest.mark.parametrize("kind", ["sgd", "adam", "adagrad", "ftrl"])
def test_fused_optimizers_on_half_tables(env, kind, vdtype):
  """configs[2]-shaped training (half rows) no longer falls back to optimizers.Generic: the fused kernels take float16 /
  bfloat16 tables (GPU value types of the reference: hkv_hashtable_op_gpu.cu.cc:1133-1138) — float32 math on the up-cast
  row and slots, ONE rounding to the storage type.  Checked against the NumPy rule (oracle/optimizers.py) applied to the
  up-cast rows: at most 1 ulp of the storage type (the float32 results of kernel and NumPy differ by <= 1e-6 relative, which
  moves a rounding only when the value sits on a tie)."""
  torch, de = env
  from oracle import optimizers as oopt
  dt = getattr(torch, vdtype)
  dim, n_keys, B = 32, 500, 4096
  opt = {"sgd": de.optimizers.SGD(0.1), "adam": de.optimizers.Adam(0.01, 0.9, 0.999, 1e-7),
         "adagrad": de.optimizers.Adagrad(0.05, 0.1), "ftrl": de.optimizers.Ftrl(0.05, -0.5, 0.1, 1e-3, 1e-3)}[kind]
  deo = de.DynamicEmbeddingOptimizer(opt)
  var = de.Variable(dim=dim, name="half_%s_%s" % (kind, vdtype), value_dtype=dt, initializer=0.25, **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  rng = np.random.default_rng(3)

  def cast(x):   # float32 -> storage type -> float32, round to nearest even like the kernel
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dt).to(torch.float32).numpy()

  ulp = 2.0 ** -10 if vdtype == "float16" else 2.0 ** -7
  init_acc = 0.1
  state = {}   # key -> [p, s1, s2] float32 arrays holding storage-representable values
  for step in range(1, 5):
    ids = rng.integers(0, n_keys, size=B).astype(np.int64)
    g = (rng.standard_normal((B, dim)) * 0.1).astype(np.float32)
    deo.apply_sparse(var, torch.from_numpy(ids).cuda(), torch.from_numpy(g).cuda())
    uniq, gsum, _ = oopt.segment_sum_by_key(ids, g)
    for k, gs in zip(uniq.tolist(), gsum):
      # a new row starts from the float32 default / initial slot values (not from their rounded images)
      p, s1, s2 = state.get(k, [np.full(dim, 0.25, np.float32), np.full(dim, init_acc if kind in ("adagrad", "ftrl") else 0.0, np.float32),
                                np.zeros(dim, np.float32)])
      if kind == "sgd":
        p = oopt.sgd(p, gs, 0.1)
      elif kind == "adam":
        p, s1, s2 = oopt.adam(p, s1, s2, gs, 0.01, 0.9, 0.999, 1e-7, step)
      elif kind == "adagrad":
        p, s1 = oopt.adagrad(p, s1, gs, 0.05)
      else:
        p, s1, s2 = oopt.ftrl(p, s1, s2, gs, 0.05, 1e-3, 1e-3)
      state[k] = [cast(p), cast(s1), cast(s2)]
  keys = np.array(sorted(state), np.int64)
  got = var.lookup(torch.from_numpy(keys).cuda()).to(torch.float32).cpu().numpy()
  want = np.stack([state[k][0] for k in keys.tolist()])
  # the duplicate sums of the fused path use a fixed tree, the oracle the sequential order: gradient sums differ by ~1e-7
  # relative, so after four steps a value may sit one storage ulp apart (two where a rounding tie flipped twice)
  err = np.abs(got - want) / np.maximum(np.abs(want), 2.0 ** -14)
  assert float(np.quantile(err, 0.999)) <= 2 * ulp and float(err.max()) <= 4 * ulp, (float(err.max()), ulp)
  assert var.size() == keys.size


def test_find_n_and_insert_n_take_the_count_on_the_device():
  """tfra_table_find_n / tfra_table_insert_or_assign_n: the chain unique -> Find -> gather -> Insert of embedding_lookup
  (python/ops/dynamic_embedding_ops.py:99-117) without the host reading the unique count in between; results identical to the
  plain entry points called with the count."""
  import torch
  import tfra_amd.dynamic_embedding as de
  from tfra_amd.dynamic_embedding import device_ops
  dim = 32
  for bounded in (False, True):
    kw = dict(init_capacity=600_000, max_capacity=600_000, evict_strategy=de.HkvEvictStrategy.LRU) if bounded else {}
    cls = de.HkvHashTable if bounded else de.CuckooHashTable
    ta = cls(torch.int64, torch.float32, torch.full((dim,), -1.0), device="cuda:0", dim=dim, name="n_a_%d" % bounded, **kw)
    tb = cls(torch.int64, torch.float32, torch.full((dim,), -1.0), device="cuda:0", dim=dim, name="n_b_%d" % bounded, **kw)
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.arange(1, 200_001, device="cuda", dtype=torch.int64) * 7919
    for t in (ta, tb):
      t._table.upsert(base, (base % 1000).to(torch.float32)[:, None].repeat(1, dim), unique_keys=True)
    for step in range(4):
      ids = base[torch.randint(0, 300, (4096,), generator=g, device="cuda") ** 2 % base.numel()]
      ids[::7] = 10**12 + step * 10_000 + torch.arange(ids[::7].numel(), device="cuda")       # never-seen ids
      uniq, idx, cnt = device_ops.unique(ids, ordered=False)
      u = int(cnt.item()) if torch.is_tensor(cnt) else int(cnt)
      ubuf = torch.full((ids.numel(),), -7, dtype=torch.int64, device="cuda")
      ubuf[:u] = uniq[:u]
      dcount = torch.tensor([u], dtype=torch.int64, device="cuda")
      vals = torch.randn((ids.numel(), dim), generator=g, device="cuda")
      # plain
      want, wex = ta.lookup(ubuf[:u], return_exists=True)
      ta._table.upsert(ubuf[:u], vals[:u], unique_keys=True)
      # device count: the buffers are as long as the batch, the tail must stay untouched
      out = torch.full((ids.numel(), dim), 123.0, device="cuda")
      got, gex = tb._table.find_n(ubuf, dcount, out=out, return_exists=True)
      tb._table.upsert_n(ubuf, dcount, vals)
      assert torch.equal(got[:u], want) and torch.equal(gex[:u], wex)
      assert bool((got[u:] == 123.0).all()) and not bool(gex[u:].any())
      assert int(ta.size().item()) == int(tb.size().item())
      a, b = ta.lookup(ubuf[:u]), tb.lookup(ubuf[:u])
      assert torch.equal(a, b) and torch.equal(a, vals[:u])
      assert not bool(tb.lookup(torch.tensor([-7], device="cuda"), return_exists=True)[1].any())   # the buffer's tail was never inserted
    # a count of zero and a count beyond the buffer
    z = torch.zeros(1, dtype=torch.int64, device="cuda")
    before = int(tb.size().item())
    tb._table.upsert_n(ubuf, z, vals)
    assert int(tb.size().item()) == before
    big = torch.tensor([10**9], dtype=torch.int64, device="cuda")
    got = tb._table.find_n(ubuf[:u], big)
    assert torch.equal(got, tb.lookup(ubuf[:u]))
    tb._table.check_errors()
