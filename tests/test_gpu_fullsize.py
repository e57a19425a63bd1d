"""GPU, BASELINE.json's FULL sizes, through size-independent properties (no oracle can hold 10^8 keys
in seconds): rows are a closed-form function of the key, so every lookup is checkable on the device.

  C2  1xMI355X: 100M keys, dim=64 fp32, Zipf-1.2 batch=131072, lookup+insert+sparse-Adam
  C3  1xMI355X: dim=128 fp16, 50 % unseen-key insert rate, bounded table (dynamic growth + eviction/export)
  C5  26 tables, mixed dim {16,32,64,128}, combined lookup + FTRL apply  (tables scaled to fit quickly)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  import tfra_amd.dynamic_embedding as de
  return torch, de


def row_of(torch, keys, dim, dtype=None):
  """row[j] = ((key * 2654435761 + j * 40503) mod 65521) / 65521 - 0.5   (exact in fp32)"""
  j = torch.arange(dim, device=keys.device, dtype=torch.int64)
  v = ((keys[:, None] * 2654435761 + j[None, :] * 40503) % 65521).to(torch.float32) / 65521.0 - 0.5
  return v if dtype is None else v.to(dtype)


def test_c2_100m_keys_dim64(env):
  torch, de = env
  from bench import keys_of_ranks_torch, zipf_bounded, keys_of_ranks
  N, dim, chunk = 100_000_000, 64, 5_000_000
  t = de.CuckooHashTable(torch.int64, torch.float32, torch.zeros(dim), init_size=int(N * 1.02), device="cuda:0", dim=dim)
  key_sum = 0
  for lo in range(1, N + 1, chunk):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(N, lo + chunk - 1) + 1, dtype=torch.int64, device="cuda"))
    t._table.upsert(k, row_of(torch, k, dim), unique_keys=True)
    key_sum = (key_sum + int(k.sum().item())) % (1 << 64)
  assert int(t.size().item()) == N
  # no growth happened: capacity was sized up front (load factor 0.75)
  assert t._table.capacity() < int(N * 1.02 / 0.75) + 100
  # Zipf lookups of resident keys + misses: every row is the closed form, misses get the default
  rng = np.random.default_rng(1)
  ids = torch.from_numpy(keys_of_ranks(zipf_bounded(rng, 131072, N))).cuda()
  out, ex = t.lookup(ids, return_exists=True)
  assert bool(ex.all())
  assert torch.equal(out, row_of(torch, ids, dim))
  miss = keys_of_ranks_torch(torch, torch.arange(N + 1, N + 100001, dtype=torch.int64, device="cuda"))
  out, ex = t.lookup(miss, dynamic_default_values=torch.full((dim,), 7.0, device="cuda"), return_exists=True)
  assert not bool(ex.any()) and bool((out == 7.0).all())
  # erase -> find -> re-insert round trip on 2M keys; size is exact at every step
  sub = keys_of_ranks_torch(torch, torch.arange(1, 2_000_001, dtype=torch.int64, device="cuda"))
  t.remove(sub)
  assert int(t.size().item()) == N - sub.numel()
  assert not bool(t.lookup(sub[:50000], return_exists=True)[1].any())
  t._table.upsert(sub, row_of(torch, sub, dim), unique_keys=True)
  assert int(t.size().item()) == N
  # idempotent upsert; checksum of all exported keys (order-free) and spot-check exported rows
  t._table.upsert(sub, row_of(torch, sub, dim), unique_keys=True)
  assert int(t.size().item()) == N
  cap = t._table.capacity()
  from tfra_amd import _capi
  from tfra_amd.dynamic_embedding.table_ops import _ptr, _stream
  win = 12_000_000
  kbuf = torch.empty(win, dtype=torch.int64, device="cuda")
  vbuf = torch.empty((win, dim), dtype=torch.float32, device="cuda")
  total, ksum = 0, 0
  for off in range(0, cap, win):
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    _capi.call("tfra_table_export_batch", t._table._h, min(win, cap - off), off, _ptr(cnt), _ptr(kbuf), _ptr(vbuf), None,
               _stream(t._table.device))
    c = int(cnt.item())
    total += c
    ksum = (ksum + int(kbuf[:c].sum().item())) % (1 << 64)
    assert torch.equal(vbuf[:1000], row_of(torch, kbuf[:1000], dim))
  assert total == N and ksum == key_sum


def test_c3_fp16_dim128_half_unseen_bounded(env):
  """50 % of every batch are never-seen keys; the table is bounded (Hkv, LRU) so it fills up and
  evicts.  Properties: size <= capacity always, and every key that IS resident returns its own row."""
  torch, de = env
  dim, cap, B = 128, 2_000_000, 131072
  t = de.HkvHashTable(torch.int64, torch.float16, torch.zeros(dim, dtype=torch.float16), init_capacity=cap,
                      max_capacity=cap, device="cuda:0", dim=dim, evict_strategy=de.HkvEvictStrategy.LRU)
  gen = torch.Generator(device="cuda").manual_seed(3)
  fresh = 1
  seen = torch.arange(1, 500001, dtype=torch.int64, device="cuda") * 7919
  t.insert(seen, row_of(torch, seen, dim, torch.float16))
  for step in range(40):
    old = seen[torch.randint(0, seen.numel(), (B // 2,), generator=gen, device="cuda")].unique()
    new = (torch.arange(fresh, fresh + B // 2, dtype=torch.int64, device="cuda") << 24) + 1  # never seen before
    fresh += B // 2
    ids = torch.cat([old, new])
    out, ex = t.lookup(ids, return_exists=True)
    want = row_of(torch, ids, dim, torch.float16)
    assert torch.equal(out[ex], want[ex])          # resident => its own row, bit exact (fp16 copy)
    assert not bool(ex[old.numel():].any())         # new keys are misses before the write-back
    t.insert(ids, want)
    n = int(t.size().item())
    assert n <= cap
  assert int(t.size().item()) > cap * 0.95          # it did fill up => eviction was exercised
  k, v = t.export()
  assert k.numel() == int(t.size().item()) and k.unique().numel() == k.numel()
  assert torch.equal(v[:20000], row_of(torch, k[:20000], dim, torch.float16))


def test_c5_26_tables_mixed_dims_ftrl(env):
  """Multi-slot DLRM shape: 26 tables, dims cycling {16,32,64,128}, one id per table per sample,
  FTRL.  Small enough for the CPU oracle: fused HIP write-back vs (1+S) finds + dense FTRL + upserts."""
  import oracle
  from oracle import optimizers as oopt
  torch, de = env
  rng = np.random.default_rng(5)
  hyper = dict(lr=0.05, l1=1e-3, l2=1e-3, init_acc=0.1)
  dims = [16, 32, 64, 128]
  B = 4096
  for ti in range(26):
    dim = dims[ti % 4]
    n_keys = int(10 ** (2 + 2.5 * ti / 25))  # log-spaced table sizes
    opt = de.optimizers.Ftrl(0.05, -0.5, 0.1, 1e-3, 1e-3)
    deo = de.DynamicEmbeddingOptimizer(opt)
    v = de.Variable(dim=dim, name="c5_%d" % ti, initializer=0.0, **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    tabs = [oracle.CpuTable(dim) for _ in range(3)]
    ora = oopt.SparseOptimizerOracle("ftrl", tabs[0], tabs[1:], hyper, 0.0)
    for step in range(2):
      ids = rng.zipf(1.2, size=B) % n_keys
      g = (rng.standard_normal((B, dim)) * 0.1).astype(np.float32)
      emb = v.lookup(torch.from_numpy(ids).cuda())
      np.testing.assert_allclose(emb.cpu().numpy(), tabs[0].find(ids, np.zeros(dim, np.float32)), rtol=2e-6, atol=2e-6)
      deo.apply_sparse(v, torch.from_numpy(ids).cuda(), torch.from_numpy(g).cuda())
      uniq, inv = np.unique(ids, return_inverse=True)
      gs = np.zeros((uniq.size, dim), np.float64); np.add.at(gs, inv, g.astype(np.float64))
      ora.apply(uniq, gs.astype(np.float32))
    k, val = v.export()
    o = np.argsort(k.cpu().numpy())
    ek, ev = tabs[0].export_sorted()
    np.testing.assert_array_equal(k.cpu().numpy()[o], ek)
    np.testing.assert_allclose(val.cpu().numpy()[o], ev, rtol=5e-6, atol=5e-6)


@pytest.mark.parametrize("cap", [4_000_000, 100_000_000])
def test_c3_1e8_slots_bench_path_properties(env, cap):
  """configs[2] at 10^8 slots through the path bench.py times: a bounded LRU table pre-filled to capacity, batches of
  131 072 ids WITH repeats = 50 % resident + 50 % never-seen, lookup then upsert_sparse (plan + single-pass ownership
  write-back with eviction).  Size-independent properties: size <= capacity, a resident key returns its own row (bit
  exact), never-seen keys miss before their write-back and hit after it, no slot is left locked, keys stay unique."""
  torch, de = env
  dim, B = 128, 131072   # cap 4 M: most keys of a batch share a home bucket with another (the left-over path carries the load)
  t = de.HkvHashTable(torch.int64, torch.float16, torch.zeros(dim, dtype=torch.float16), init_capacity=cap, max_capacity=cap,
                      device="cuda:0", dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="c3_%d" % cap)
  for lo in range(1, cap + 1, 4_000_000):
    k = torch.arange(lo, min(cap, lo + 3_999_999) + 1, dtype=torch.int64, device="cuda") * 7919
    t._table.upsert(k, row_of(torch, k, dim, torch.float16), unique_keys=True)
  n0 = int(t.size().item())
  assert 0.9 * cap < n0 <= cap
  gen = torch.Generator(device="cuda").manual_seed(9)
  fresh = 1
  for step in range(12):
    old = torch.randint(1, cap + 1, (B // 2,), generator=gen, device="cuda") * 7919
    old[: B // 8] = old[0]                              # a hot id: thousands of repeats in the batch
    new = -(torch.arange(fresh, fresh + B // 2, dtype=torch.int64, device="cuda") * 104729) - 5   # never seen before (negative)
    fresh += B // 2
    ids = torch.cat([old, new])[torch.randperm(B, generator=gen, device="cuda")]
    out, ex = t.lookup(ids, return_exists=True)
    want = row_of(torch, ids, dim, torch.float16)
    assert torch.equal(out[ex], want[ex])
    assert not bool(ex[ids < 0].any())
    t._table.upsert_sparse(ids, want)
    out2, ex2 = t.lookup(ids, return_exists=True)
    assert float(ex2.float().mean()) > 0.999             # written back (LRU: the newest entries are not the victims)
    assert torch.equal(out2[ex2], want[ex2])
    assert int(t.size().item()) <= cap
  c = t._table.slot_census()
  assert c["locked"] == 0 and c["live"] == int(t.size().item())
  t._table.check_errors()
  kbuf = torch.empty(4_000_000, dtype=torch.int64, device="cuda")
  vbuf = torch.empty((4_000_000, dim), dtype=torch.float16, device="cuda")
  cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
  from tfra_amd import _capi
  from tfra_amd.dynamic_embedding.table_ops import _ptr, _stream
  _capi.call("tfra_table_export_batch", t._table._h, 3_000_000, cap // 4, _ptr(cnt), _ptr(kbuf), _ptr(vbuf), None, _stream(t._table.device))
  m = int(cnt.item())
  assert m > 2_000_000 and kbuf[:m].unique().numel() == m
  assert torch.equal(vbuf[:50_000], row_of(torch, kbuf[:50_000], dim, torch.float16))


def test_metric_config_1e9_slots_benchmarked_paths(env):
  """BASELINE.json's metric configuration at its FULL size — a 10^9-slot bounded LRU table, dim 64 fp32 (273 GB of the
  288 GB) — through the three paths bench.py times: the look-ahead step driver (tfra_table_step_prefetch_assign), the plain
  calls (find + upsert_sparse) and the reference's op surface (find -> unique -> insert_or_assign with unique keys, the
  ownership pass fed directly).  Rows are a closed form of (key, version), so every lookup is checkable on the device:
  size <= capacity; a resident key returns its own row bit-exactly; never-seen ids miss before their write-back and hit
  after it; repeats: the LAST occurrence wins; no slot stays locked; check_errors is clean; exported keys are unique."""
  torch, de = env
  from bench import keys_of_ranks_torch, IdFactory
  dim, B, cap = 64, 131072, 1_000_000_000
  try:
    t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=cap, max_capacity=cap, device="cuda:0", dim=dim,
                        evict_strategy=de.HkvEvictStrategy.LRU, name="m1b_full")
  except Exception as e:   # a box with less free HBM than the benchmark needs
    pytest.skip("10^9 slots do not allocate here: %s" % str(e)[:120])
  tbl = t._table
  capacity = tbl.capacity()
  for lo in range(1, cap + 1, 4_000_000):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(cap, lo + 3_999_999) + 1, dtype=torch.int64, device="cuda"))
    tbl.upsert(k, row_of(torch, k, dim), unique_keys=True)
  n0 = int(t.size().item())
  assert 0.9 * cap < n0 <= capacity

  def versioned(keys, ver):   # the row a key gets when it is written in step `ver`
    return row_of(torch, keys * 31 + ver, dim)

  # resident keys return the row of their pre-fill (those the pre-fill's own evictions removed miss)
  idf = IdFactory(torch, torch.device("cuda", 0), B, cap, 0.25, cap + 1, 1234)   # 25 % never-seen ids per batch
  ids = idf.keys(1)[0]
  out, ex = t.lookup(ids, return_exists=True)
  assert torch.equal(out[ex], row_of(torch, ids, dim)[ex])
  assert not bool(ex[idf.pos].any())            # the never-seen quarter of the batch misses
  # (the pre-fill's own evictions took the OLDEST entries = the lowest ranks = the hottest ids of the Zipf stream: a good part
  # of the remaining positions misses too until the first steps have written those ids back)
  assert float(ex.float().mean()) > 0.2
  written = {}   # key tensor -> version, for a final spot check

  # (1) look-ahead driver: 6 steps, values differ per step AND per position (repeats: the last one wins)
  batches = idf.keys(7)
  ps = de.PrefetchAssignStep(t).prime(batches[0])
  for s in range(6):
    b = batches[s]
    pos = torch.arange(B, device="cuda")
    vals = versioned(b, 100 + s) + (pos[:, None] % 7).to(torch.float32)     # position-dependent: tells WHICH occurrence was kept
    fresh = b[~t.lookup(b, return_exists=True)[1]]
    ps.step(vals, batches[s + 1])
    got, ex = t.lookup(b, return_exists=True)
    assert bool(ex.all())                                                   # written back (never-seen ids included)
    uk, inv = torch.unique(b, return_inverse=True)
    lp = torch.zeros(uk.numel(), dtype=torch.long, device="cuda")
    lp.scatter_reduce_(0, inv, pos, reduce="amax", include_self=False)
    assert torch.equal(got, vals[lp][inv])
    assert fresh.numel() > B // 8
    assert int(t.size().item()) <= capacity
  del ps
  # (2) plain calls
  for s in range(4):
    b = idf.keys(1)[0]
    vals = versioned(b, 200 + s)
    out, ex = t.lookup(b, return_exists=True)
    tbl.upsert_sparse(b, vals)
    got, ex2 = t.lookup(b, return_exists=True)
    assert bool(ex2.all()) and torch.equal(got, vals)
  # (3) op surface: unique keys straight into insert_or_assign (TFRA_FLAG_UNIQUE_KEYS)
  for s in range(4):
    b = idf.keys(1)[0]
    u, _, _ = de.device_ops.unique(b)
    vals = versioned(u, 300 + s)
    miss_before = ~t.lookup(u, return_exists=True)[1]
    tbl.upsert(u, vals, unique_keys=True)
    got, ex2 = t.lookup(u, return_exists=True)
    assert bool(ex2.all()) and torch.equal(got, vals)
    assert int(miss_before.sum().item()) > u.numel() // 8                    # never-seen ids did miss before
    written[s] = (u[:4096].clone(), vals[:4096].clone())
  for s, (k, v) in written.items():
    if s == 3:   # the last call's keys are still there (earlier ones may have been overwritten by later batches' hot ids)
      got, ex = t.lookup(k, return_exists=True)
      assert bool(ex.all()) and torch.equal(got, v)
  c = tbl.slot_census()
  assert c["locked"] == 0 and c["live"] == int(t.size().item()) <= capacity
  tbl.check_errors()
  kbuf = torch.empty(2_000_000, dtype=torch.int64, device="cuda")
  vbuf = torch.empty((2_000_000, dim), dtype=torch.float32, device="cuda")
  cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
  from tfra_amd import _capi
  from tfra_amd.dynamic_embedding.table_ops import _ptr, _stream
  _capi.call("tfra_table_export_batch", tbl._h, 1_500_000, capacity // 3, _ptr(cnt), _ptr(kbuf), _ptr(vbuf), None, _stream(tbl.device))
  m = int(cnt.item())
  assert m > 1_000_000 and kbuf[:m].unique().numel() == m
  del t, tbl, kbuf, vbuf
  import gc
  gc.collect()
  torch.cuda.empty_cache()


def _full_table(torch, de, dim, dtype, cap, name):
  try:
    t = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=cap, max_capacity=cap, device="cuda:0", dim=dim,
                        evict_strategy=de.HkvEvictStrategy.LRU, name=name)
  except Exception as e:   # a box with less free HBM than the benchmark needs
    pytest.skip("%d slots do not allocate here: %s" % (cap, str(e)[:120]))
  from bench import keys_of_ranks_torch
  for lo in range(1, cap + 1, 4_000_000):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(cap, lo + 3_999_999) + 1, dtype=torch.int64, device="cuda"))
    t._table.upsert(k, row_of(torch, k, dim, dtype), unique_keys=True)
  return t


def _release(torch, *objs):
  import gc
  for o in objs:
    del o
  gc.collect()
  torch.cuda.empty_cache()


@pytest.mark.parametrize("cfg", ["metric_dim64_f32", "configs2_dim128_f16"])
def test_overlapped_step_1e9_slots(env, cfg):
  """The driver behind bench.py's `value` (tfra_table_step_overlap: lookup of batch i+1, write-back of batch i, its tail and the
  plan builders in ONE launch) at the metric's FULL size — 10^9 slots, 273 GB — and at configs[2]'s (dim 128 fp16, half of every
  batch never-seen ids: an eviction per new key).  Per step: the lookup must equal a plain tfra_table_find issued right behind the
  call (the table then holds exactly what this lookup had to reflect: module docstring of tests/test_gpu_overlap.py), resident keys
  return their closed-form row, never-seen ids miss.  The batches are drawn so that the write-back's victims — the oldest
  entries — ARE ids of the next lookup (ranks of the first pre-fill chunks): the deferred-eviction / correction path runs at size."""
  torch, de = env
  from bench import keys_of_ranks_torch
  dim, dtype = (64, torch.float32) if cfg.startswith("metric") else (128, torch.float16)
  B, cap, steps = 131072, 1_000_000_000, 10
  t = _full_table(torch, de, dim, dtype, cap, "ovl_full_%s" % cfg)
  tbl = t._table
  capacity = tbl.capacity()
  assert 0.9 * cap < int(t.size().item()) <= capacity
  gen = torch.Generator(device="cuda").manual_seed(77)
  fresh = cap + 1

  def batch(step):
    nonlocal fresh
    # a third Zipf-ish hot ranks, a third of the OLDEST surviving ranks (the pre-fill went from rank 1 up and evicted the lowest ~5 %
    # itself: ranks 40 M .. 120 M straddle that edge — some absent, the others the least recently used entries), a third never seen
    hot = (torch.rand(B // 3, generator=gen, device="cuda") ** 8 * cap).to(torch.int64) + 1
    old = torch.randint(40_000_000, 120_000_000, (B // 3,), generator=gen, device="cuda")
    new = torch.arange(fresh, fresh + B - 2 * (B // 3), dtype=torch.int64, device="cuda")
    fresh += new.numel()
    r = torch.cat([hot, old, new])[torch.randperm(B, generator=gen, device="cuda")]
    return keys_of_ranks_torch(torch, r), r > cap

  ids = [batch(s) for s in range(steps + 2)]
  drv = de.OverlapAssignStep(t).prime(ids[0][0])
  written = {}
  # an INDEPENDENT model of the row contents across steps (not the table itself): a sorted dictionary on the device of every key the
  # steps have written with the row of its last write; a key no step wrote holds its closed-form pre-fill row
  seen_k = torch.empty(0, dtype=torch.int64, device="cuda")
  seen_v = torch.empty((0, dim), dtype=dtype, device="cuda")
  n_from_dict = 0
  for s in range(steps):
    k, is_new = ids[s]
    vals = row_of(torch, k * 31 + (s + 1), dim, dtype) + (torch.arange(B, device="cuda")[:, None] % 5).to(dtype)   # position-dependent: WHICH occurrence is kept
    out, ex = drv.step(vals, ids[s + 1][0], ids[s + 2][0], return_exists=True)
    ref, rex = tbl.find(k, return_exists=True)
    assert torch.equal(ex, rex), "step %d: %d exists flags differ" % (s, int((ex != rex).sum()))
    assert torch.equal(out, ref), "step %d" % s
    assert not bool(ex[is_new].any())                    # never-seen ids miss (a never-seen rank is drawn once)
    assert int(t.size().item()) <= capacity
    want = row_of(torch, k, dim, dtype)                  # the pre-fill's row of the key ...
    if seen_k.numel():
      pos = torch.searchsorted(seen_k, k).clamp(max=seen_k.numel() - 1)
      hit = seen_k[pos] == k
      want = torch.where(hit[:, None], seen_v[pos], want)   # ... unless an earlier step wrote it: the row of its last write
      n_from_dict += int((hit & ex).sum())
    assert torch.equal(out[ex], want[ex]), "step %d: a present key does not hold the row of its last write" % s
    assert bool((out[~ex] == 0).all()), "step %d: an absent key did not read the default row" % s
    uk, inv = torch.unique(k, return_inverse=True)
    lp = torch.zeros(uk.numel(), dtype=torch.long, device="cuda")
    lp.scatter_reduce_(0, inv, torch.arange(B, device="cuda"), reduce="amax", include_self=False)
    allk, allv = torch.cat([seen_k, uk]), torch.cat([seen_v, vals[lp]])
    seen_k, inv2 = torch.unique(allk, return_inverse=True)
    last = torch.zeros(seen_k.numel(), dtype=torch.long, device="cuda")
    last.scatter_reduce_(0, inv2, torch.arange(allk.numel(), device="cuda"), reduce="amax", include_self=False)   # the newer entry wins
    seen_v = allv[last]
    written[s] = (k, vals)
  assert n_from_dict > steps * 1000                      # the hot ids recur: thousands of lookups per step were checked against earlier writes
  drv.flush()
  st = drv.stats()
  assert st["overlapped"] >= steps - 3 and st["deferred_evictions"] > 0 and st["rows_corrected"] > 0, st
  assert st["plans_built_in_launch"] >= steps - 4, st
  # the last batch is in the table, last occurrence wins
  k, vals = written[steps - 1]
  got, ex = t.lookup(k, return_exists=True)
  uk, inv = torch.unique(k, return_inverse=True)
  lp = torch.zeros(uk.numel(), dtype=torch.long, device="cuda")
  lp.scatter_reduce_(0, inv, torch.arange(B, device="cuda"), reduce="amax", include_self=False)
  assert bool(ex.all()) and torch.equal(got, vals[lp][inv])
  c = tbl.slot_census()
  assert c["locked"] == 0 and c["live"] == int(t.size().item()) <= capacity
  tbl.check_errors()
  _release(torch, drv, t, tbl, ids, written)


def test_configs3_5e8_keys_routed_step_world1(env):
  """configs[3]'s per-GPU workload at its full size: 5*10^8 resident keys behind the route driver (tfra_route_*, world 1: device
  copies where the alltoalls would be), per-GPU batch 131 072 from a Zipf-like stream, fused SGD.  Closed-form rows: a lookup
  returns row(key) for a key that has not been trained yet and row(key) - lr * (sum of its gradients) afterwards."""
  torch, de = env
  from bench import keys_of_ranks_torch
  from tfra_amd.dynamic_embedding.distributed import NativeRoutedStep
  dim, B, n_keys, lr = 64, 131072, 500_000_000, 0.5
  opt = de.optimizers.SGD(lr)
  deo = de.DynamicEmbeddingOptimizer(opt)
  try:
    var = de.Variable(dim=dim, devices=["cuda:0"], name="c4_full", initializer=0.0, init_size=int(n_keys * 1.05) + (1 << 20),
                      **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    table = var.tables[0]
    for lo in range(1, n_keys + 1, 4_000_000):
      k = keys_of_ranks_torch(torch, torch.arange(lo, min(n_keys, lo + 3_999_999) + 1, dtype=torch.int64, device="cuda"))
      table._table.upsert(k, row_of(torch, k, dim), unique_keys=True)
  except Exception as e:
    pytest.skip("5*10^8 keys do not fit here: %s" % str(e)[:120])
  assert int(table.size().item()) == n_keys
  gen = torch.Generator(device="cuda").manual_seed(5)
  steps, ahead = 6, 3
  ids = [keys_of_ranks_torch(torch, (torch.rand(B, generator=gen, device="cuda") ** 6 * n_keys).to(torch.int64) + 1) for _ in range(steps + ahead)]
  rs = NativeRoutedStep(var, deo, partition_mode=0, max_batch=B)
  for j in range(ahead):
    rs.feed(ids[j])
  gsum = {}   # key -> summed gradient so far, on the device: a dense accumulator over the ids seen
  seen_k = torch.empty(0, dtype=torch.int64, device="cuda")
  seen_g = torch.empty((0, dim), device="cuda")
  for s in range(steps):
    out = rs.lookup()
    k = ids[s]
    want = row_of(torch, k, dim)
    if seen_k.numel():
      pos = torch.searchsorted(seen_k, k).clamp(max=seen_k.numel() - 1)
      hit = seen_k[pos] == k
      want = torch.where(hit[:, None], want - lr * seen_g[pos], want)
    # (the reference sum here is torch's index_add_ — atomics, any order — not the sequential oracle of the 1e-6 parity tests:
    # a hot id's thousands of gradients sum to ~1, two fp32 summation orders of that differ by ~1e-6 of it)
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    g = torch.randn((B, dim), generator=gen, device="cuda") * 0.01
    rs.apply(g)
    uk, inv = torch.unique(k, return_inverse=True)
    ug = torch.zeros((uk.numel(), dim), device="cuda").index_add_(0, inv, g)
    allk = torch.cat([seen_k, uk])
    allg = torch.cat([seen_g, ug])
    seen_k, inv2 = torch.unique(allk, return_inverse=True)
    seen_g = torch.zeros((seen_k.numel(), dim), device="cuda").index_add_(0, inv2, allg)
    rs.feed(ids[s + ahead])
  for _ in range(ahead):
    rs.lookup(); rs.apply(torch.zeros((B, dim), device="cuda"))
  torch.cuda.synchronize()
  rs.close()
  assert int(table.size().item()) == n_keys
  table._table.check_errors()
  _release(torch, rs, var, table, ids)
