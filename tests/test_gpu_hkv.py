"""GPU: HkvHashTable flavour — per-key scores and in-bucket min-score eviction of a bounded table.
HierarchicalKV itself is third party and not on disk; these are the
behavioural KATs the reference's own GPU tests hold at that boundary
(K14 = T/hkv_hashtable_evict_test.py:241-573, K15 = T/hkv_hashtable_ops_test.py:572-625)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DIM = 8


@pytest.fixture(scope="module")
def env():
  import torch
  import tfra_amd.dynamic_embedding as de
  return torch, de


def make(de, torch, strategy, name, **cfg):
  return de.get_variable(name, key_dtype=torch.int64, value_dtype=torch.int32, initializer=0, dim=DIM, init_size=1024,
                         kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(
                             init_capacity=1024, max_capacity=1024, max_hbm_for_values=1024 * 64,
                             evict_strategy=strategy, **cfg)))


def up(torch, table, keys, val):
  k = torch.from_numpy(np.asarray(keys, np.int64)).cuda()
  v = torch.full((k.numel(), DIM), val, dtype=torch.int32, device="cuda")
  table.upsert(k, v)


def scores_of(table):
  k, s = table.tables[0].export_keys_and_scores(1)
  return k.cpu().numpy(), s.cpu().numpy()


def test_k14_lfu(env):
  """T/hkv_hashtable_evict_test.py:241-310"""
  torch, de = env
  t = make(de, torch, de.HkvEvictStrategy.LFU, "k14_lfu")
  up(torch, t, [0, 1, 2, 3], 1)
  np.testing.assert_array_equal(scores_of(t)[1], np.ones(4))
  up(torch, t, [0, 1, 2, 3], 1)
  np.testing.assert_array_equal(scores_of(t)[1], np.full(4, 2))
  up(torch, t, [0, 1, 4, 5], 1)
  np.testing.assert_array_equal(np.sort(scores_of(t)[1]), [1, 1, 2, 2, 3, 3])
  up(torch, t, np.arange(4, 1034), 10)
  k, s = scores_of(t)
  assert len(k) < 1024 and len(k) == int(t.size().item())
  np.testing.assert_array_equal(np.sort(k)[:6], np.arange(6))          # the frequent keys survive
  np.testing.assert_array_equal(np.sort(s)[-6:], [2, 2, 2, 2, 3, 3])
  assert len(np.unique(k)) == len(k)


def test_k14_epoch_lfu(env):
  """T/hkv_hashtable_evict_test.py:312-406: score = (epoch << 32) + count, epoch steps every 4 upserts."""
  torch, de = env
  t = make(de, torch, de.HkvEvictStrategy.EPOCHLFU, "k14_elfu", step_per_epoch=4)
  for base in (1, 1 + (1 << 32), 1 + (2 << 32)):
    up(torch, t, [0, 1, 2, 3], 1)
    assert np.all(np.sort(scores_of(t)[1])[-4:] >= base)
    up(torch, t, [0, 1, 2, 3], 1)
    assert np.all(np.sort(scores_of(t)[1])[-4:] >= base + 1)
    up(torch, t, [0, 1, 4, 5], 1)
    assert np.all(np.sort(scores_of(t)[1])[-6:] >= np.array([base, base, base + 1, base + 1, base + 2, base + 2]))
    up(torch, t, np.arange(4, 1024), 10)
    k, s = scores_of(t)
    assert len(k) < 1024
    np.testing.assert_array_equal(np.sort(k)[:6], np.arange(6))
    assert np.all(np.sort(s)[-6:] >= np.array([base + 1] * 4 + [base + 2] * 2))


def test_k14_lru(env):
  """T/hkv_hashtable_evict_test.py:408-479: later upsert => larger score; the oldest keys go first."""
  torch, de = env
  t = make(de, torch, de.HkvEvictStrategy.LRU, "k14_lru")
  up(torch, t, [0, 1, 2, 3], 1)
  assert np.all(np.isin([0, 1, 2, 3], scores_of(t)[0]))
  up(torch, t, [2, 3, 6, 7], 1)
  k, s = scores_of(t)
  d = dict(zip(k.tolist(), s.tolist()))
  assert max(d[0], d[1]) < min(d[2], d[3])
  up(torch, t, np.arange(4, 1044), 10)
  up(torch, t, np.arange(1024, 1400), 10)
  k, s = scores_of(t)
  assert len(k) <= 1024 and len(np.unique(k)) == len(k)
  assert not np.any(np.isin([0, 1, 2, 3], k))


def test_k14_customized(env):
  """T/hkv_hashtable_evict_test.py:527-573: a batch with score 10000 survives a batch with score 1."""
  torch, de = env

  def gen(keys):
    return torch.where(keys >= 2048, torch.full_like(keys, 10000), torch.ones_like(keys))

  t = make(de, torch, de.HkvEvictStrategy.CUSTOMIZED, "k14_custom", gen_scores_fn=gen)
  up(torch, t, np.arange(2048, 4096), 10)
  up(torch, t, np.arange(0, 1024), 10)
  k, s = scores_of(t)
  assert len(k) > 900
  assert np.all(s == 10000) and np.all(k >= 1024)
  # evicting never corrupts rows: every resident key still maps to its row
  out = t.lookup(torch.from_numpy(k).cuda()).cpu().numpy()
  assert np.all(out == 10)


def test_k15_repeat_insert_and_bounded_size(env):
  """T/hkv_hashtable_ops_test.py:572-625: 50 000 keys into capacity 100 000, twice: size == 50 000 both
  times (nothing may be evicted at load factor 0.5); :627-686: overflow keeps size in [N/2, N]."""
  torch, de = env
  for dim, dt in [(1, torch.int8), (8, torch.int32), (10, torch.int64), (64, torch.float32), (200, torch.float16)]:
    t = de.get_variable("k15_%d" % dim, key_dtype=torch.int64, value_dtype=dt, initializer=-1, dim=dim,
                        kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(init_capacity=100000,
                                                                                       max_capacity=100000)))
    keys = torch.arange(50000, dtype=torch.int64).cuda()
    vals = (keys % 100)[:, None].repeat(1, dim).to(dt)
    assert int(t.size().item()) == 0
    for i in range(2):
      t.upsert(keys, vals)
      assert int(t.size().item()) == 50000
    np.testing.assert_array_equal(t.lookup(keys).cpu().numpy(), vals.cpu().numpy())
    t.clear()
    assert int(t.size().item()) == 0
  t = de.get_variable("k15_over", key_dtype=torch.int64, value_dtype=torch.float32, initializer=0.0, dim=4,
                      kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(init_capacity=65536,
                                                                                     max_capacity=65536)))
  t.upsert(torch.arange(200000, dtype=torch.int64).cuda(), torch.ones((200000, 4)).cuda())
  n2 = int(t.size().item())
  assert 65536 // 2 <= n2 <= 65536
  k, v = t.export()
  assert len(np.unique(k.cpu().numpy())) == n2


# ---- fused optimizer write-back on a bounded table that is full ------------------------------------
def _opt_pair(de, kind):
  o = de.optimizers
  if kind == "sgd":
    return o.SGD(0.1), dict(lr=0.1)
  if kind == "adam":
    return o.Adam(0.01, 0.9, 0.999, 1e-8), dict(lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8)
  if kind == "adagrad":
    return o.Adagrad(0.05, 0.1), dict(lr=0.05, init_acc=0.1)
  return o.Ftrl(0.05, l1_regularization_strength=1e-3, l2_regularization_strength=1e-3), dict(lr=0.05, l1=1e-3, l2=1e-3)


@pytest.mark.parametrize("dim", [8, 6])          # 8: tfra_table_apply_sparse; 6: unique + segment_sum + apply_optimizer
@pytest.mark.parametrize("kind", ["sgd", "adam", "adagrad", "ftrl"])
def test_fused_optimizer_evicts_on_full_bounded_table(env, kind, dim):
  """The reference trains an HkvHashTable at capacity with find (miss -> default) + dense apply + upsert,
  and the upsert evicts (PY/dynamic_embedding_optimizer.py:165-204, lookup_table_op_hkv.h:522-537).  The
  fused write-back must do the same: new keys replace min-score entries instead of failing."""
  torch, de = env
  import oracle
  from oracle import optimizers as oopt
  rng = np.random.default_rng(dim * 10 + len(kind))
  opt, hyper = _opt_pair(de, kind)
  deo = de.DynamicEmbeddingOptimizer(opt)
  var = de.get_variable("hkv_fused_%s_%d" % (kind, dim), key_dtype=torch.int64, value_dtype=torch.float32,
                        initializer=0.25, dim=dim, init_size=1024,
                        kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(
                            init_capacity=1024, max_capacity=1024, max_hbm_for_values=1 << 20,
                            evict_strategy=de.HkvEvictStrategy.LRU)),
                        **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  for lo in range(0, 3000, 500):   # fill beyond capacity: the table is now as full as it gets
    k = np.arange(lo, lo + 500, dtype=np.int64)
    var.upsert(torch.from_numpy(k).cuda(), torch.from_numpy(np.tile(k[:, None] * 1e-3, (1, dim)).astype(np.float32)).cuda())
  n0 = int(var.size())
  assert 900 < n0 <= 1024
  rk, rv = var.export()
  rk, rv = rk.cpu().numpy(), rv.cpu().numpy()
  # CPU model of the reference sequence, seeded with what is resident
  tabs = [oracle.CpuTable(dim) for _ in range(1 + len(opt.slots))]
  tabs[0].insert(rk, rv)
  ora = oopt.SparseOptimizerOracle(kind, tabs[0], tabs[1:], hyper, 0.25)
  fresh_base = 10**6
  for step in range(4):
    resident = rng.choice(rk, size=100, replace=False)
    fresh = np.arange(fresh_base, fresh_base + 150, dtype=np.int64)
    fresh_base += 150
    ids = np.concatenate([resident, resident[:40], fresh, fresh[:30]])
    rng.shuffle(ids)
    g = rng.standard_normal((ids.size, dim)).astype(np.float32)
    # the oracle must not know keys the engine has evicted meanwhile: drop them from the model first
    live, _ = var.export()
    live = set(live.cpu().numpy().tolist())
    for t in tabs:
      ks, _ = t.export_sorted()
      gone = np.array([x for x in ks.tolist() if x not in live], dtype=np.int64)
      if gone.size:
        t.remove(gone)
    uniq = ora.apply(ids, g)
    deo.apply_sparse(var, torch.from_numpy(ids).cuda(), torch.from_numpy(g).cuda())
    size = int(var.size())                      # raises if any key could neither be placed nor evict
    assert size <= 1024
    got, ex = var.lookup(torch.from_numpy(uniq).cuda(), return_exists=True)
    missing = uniq[~ex.cpu().numpy()]
    assert missing.size == 0, "step %d: keys of the step not resident after its write-back: %s (fresh: %s)" % (
        step, missing[:10], (missing >= 10**6)[:10])
    want = tabs[0].find(uniq, np.full(dim, 0.25, np.float32))
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-6, atol=2e-6)
    rk = var.export()[0].cpu().numpy()
  # nothing was duplicated or lost track of
  k, _ = var.export()
  k = k.cpu().numpy()
  assert len(np.unique(k)) == len(k) == int(var.size())


def test_fused_optimizer_lfu_admission_on_full_table(env):
  """LFU: a new key (count 1) only replaces an entry whose count is not higher; frequent keys survive."""
  torch, de = env
  opt = de.optimizers.SGD(0.1)
  deo = de.DynamicEmbeddingOptimizer(opt)
  var = de.get_variable("hkv_fused_lfu", key_dtype=torch.int64, value_dtype=torch.float32, initializer=0.0, dim=8,
                        init_size=1024, kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(
                            init_capacity=1024, max_capacity=1024, max_hbm_for_values=1 << 20,
                            evict_strategy=de.HkvEvictStrategy.LFU)))
  hot = torch.arange(0, 64, device="cuda")
  g1 = torch.ones((64, 8), device="cuda")
  for _ in range(5):
    deo.apply_sparse(var, hot, g1)               # count 5 each
  for lo in range(1000, 6000, 500):             # 5000 one-shot keys through a 1020-slot table
    ids = torch.arange(lo, lo + 500, device="cuda")
    deo.apply_sparse(var, ids, torch.ones((500, 8), device="cuda"))
    assert int(var.size()) <= 1024
  v, ex = var.lookup(hot, return_exists=True)
  assert bool(ex.all())
  np.testing.assert_allclose(v.cpu().numpy(), np.full((64, 8), -0.5, np.float32), rtol=1e-6)


def test_accum_evicts_on_full_bounded_table(env):
  """accum_or_assign on a full bounded table (the bp_v2 write-back, PY/dynamic_embedding_variable.py:806-855):
  (absent, exists=False) inserts by evicting a minimum-score entry; (present, exists=True) adds in place;
  mismatching flags stay no-ops (K/lookup_impl/lookup_table_op_cpu.h accumrase semantics)."""
  torch, de = env
  rng = np.random.default_rng(9)
  t = de.get_variable("hkv_accum_full", key_dtype=torch.int64, value_dtype=torch.float32, initializer=0.0, dim=DIM,
                      init_size=1024, kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(
                          init_capacity=1024, max_capacity=1024, max_hbm_for_values=1 << 20,
                          evict_strategy=de.HkvEvictStrategy.LRU)))
  for lo in range(0, 3000, 500):
    k = np.arange(lo, lo + 500, dtype=np.int64)
    t.upsert(torch.from_numpy(k).cuda(), torch.from_numpy(np.tile(k[:, None] * 1.0, (1, DIM)).astype(np.float32)).cuda())
  assert int(t.size()) >= 1000
  for step in range(4):
    rk, rv = t.export()
    rk, rv = rk.cpu().numpy(), rv.cpu().numpy()
    pick = rng.choice(rk.size, size=120, replace=False)
    present_add, present_noop = rk[pick[:80]], rk[pick[80:]]
    fresh_ins = np.arange(10**7 + step * 300, 10**7 + step * 300 + 200, dtype=np.int64)
    absent_noop = np.arange(2 * 10**7 + step * 50, 2 * 10**7 + step * 50 + 50, dtype=np.int64)
    keys = np.concatenate([present_add, present_noop, fresh_ins, absent_noop])
    exists = np.concatenate([np.ones(80, bool), np.zeros(40, bool), np.zeros(200, bool), np.ones(50, bool)])
    order = rng.permutation(keys.size)
    keys, exists = keys[order], exists[order]
    delta = rng.integers(-8, 8, size=(keys.size, DIM)).astype(np.float32)
    old = np.zeros_like(delta)
    # Variable.accum computes where(exists, new - old, new): feed (old=0, new=delta)
    t.accum(torch.from_numpy(keys).cuda(), torch.from_numpy(old).cuda(), torch.from_numpy(delta).cuda(),
            torch.from_numpy(exists).cuda())
    assert int(t.size()) <= 1024               # raises if a key could neither be placed nor evict
    got, ex = t.lookup(torch.from_numpy(keys).cuda(), return_exists=True)
    got, ex = got.cpu().numpy(), ex.cpu().numpy()
    row_of = dict(zip(rk.tolist(), rv))
    fresh_set, noop_set = set(fresh_ins.tolist()), set(absent_noop.tolist())
    for i, (k, e) in enumerate(zip(keys.tolist(), exists.tolist())):
      if k in noop_set:                         # absent & exists: dropped
        assert not ex[i]
      elif k in fresh_set:                      # absent & !exists: inserted (evicting), row = the delta
        assert ex[i], "fresh key %d of step %d not resident" % (k, step)
        np.testing.assert_array_equal(got[i], delta[i])
      elif ex[i]:                               # still resident (LRU keeps what this step touched)
        want = row_of[k] + delta[i] if e else row_of[k]
        np.testing.assert_array_equal(got[i], want)
      else:                                     # only an untouched (no-op) resident may have been evicted
        assert not e


# ---- the reference's per-strategy smoke cases (T/hkv_hashtable_evict_test.py:110-239) ----------------
def _gen_scores_plus_one(keys):
  return keys + 1          # gen_scores_fn of the reference tests (:88-91)


@pytest.mark.parametrize("strategy_name", ["LRU", "LFU", "EPOCHLRU", "EPOCHLFU", "CUSTOMIZED"])
def test_evict_strategy_basic_and_score_exports(env, strategy_name):
  """test_evict_strategy (:110-153), test_export_keys_and_scores (:155-196), test_export_with_scores (:198-239):
  for every strategy upsert 4 keys, read them back, export (keys, scores) and (keys, values, scores)."""
  torch, de = env
  strategy = getattr(de.HkvEvictStrategy, strategy_name)
  t = de.get_variable("evs_" + strategy_name, key_dtype=torch.int64, value_dtype=torch.int32, initializer=0, dim=DIM,
                      init_size=1024, kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(
                          init_capacity=1024, max_capacity=1024, max_hbm_for_values=1024 * 64,
                          evict_strategy=strategy, gen_scores_fn=_gen_scores_plus_one)))
  assert int(t.size()) == 0
  keys = torch.tensor([0, 1, 2, 3], device="cuda")
  values = torch.tensor([[0] * DIM, [1] * DIM, [2] * DIM, [3] * DIM], dtype=torch.int32, device="cuda")
  t.upsert(keys, values)
  assert torch.equal(t.lookup(keys), values)
  ek, es = t.tables[0].export_keys_and_scores(1)
  ek, es = ek.cpu().numpy(), es.cpu().numpy()
  np.testing.assert_array_equal(np.sort(ek), [0, 1, 2, 3])
  k2, v2, s2 = t.tables[0].export_with_scores(1)
  k2, v2, s2 = k2.cpu().numpy(), v2.cpu().numpy(), s2.cpu().numpy()
  o = np.argsort(k2)
  np.testing.assert_array_equal(k2[o], [0, 1, 2, 3])
  np.testing.assert_array_equal(v2[o], values.cpu().numpy())          # export order is unspecified: compare by key
  for keys_x, scores_x in ((ek, es), (k2, s2)):
    if strategy_name == "CUSTOMIZED":
      np.testing.assert_array_equal(scores_x[np.argsort(keys_x)], [1, 2, 3, 4])
    elif strategy_name in ("LFU", "EPOCHLFU"):
      np.testing.assert_array_equal(scores_x, np.ones(4))
    else:
      assert (scores_x > 0).all()                                     # a clock value
  with pytest.raises(ValueError):
    t.tables[0].export_keys_and_scores(0)                             # split_size must be a positive integer


def test_reach_max_hbm(env):
  """T/hkv_hashtable_ops_test.py:627-693: 2 Mi-slot table, int64 values dim 32; the first half of the keys all
  fit (size == n/2 exactly: nothing is evicted at load factor 0.5), the second half fills/evicts within bounds."""
  torch, de = env
  n, dim = 1024 * 1024 * 2, 32
  t = de.get_variable("reach_max_hbm", key_dtype=torch.int64, value_dtype=torch.int64,
                      initializer=np.array([-1], dtype=np.int64), dim=dim,
                      kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(
                          init_capacity=n, max_capacity=n, max_hbm_for_values=8 * (dim + 1) * 1024 * 1024 * 4)))
  t.clear()
  assert int(t.size()) == 0
  k = torch.arange(0, n // 2, device="cuda")
  t.upsert(k, k[:, None].repeat(1, dim))
  assert int(t.size()) == n // 2
  k2 = torch.arange(n // 2, n, device="cuda")
  t.upsert(k2, k2[:, None].repeat(1, dim))
  assert n // 2 <= int(t.size()) <= n
  probe = torch.arange(n - 1000, n, device="cuda")          # the newest keys are resident (LRU) with their rows
  got, ex = t.lookup(probe, return_exists=True)
  assert bool(ex.all()) and torch.equal(got, probe[:, None].repeat(1, dim))
  t.clear()
  assert int(t.size()) == 0


def test_basic_odd_capacity_and_reserved_bit(env):
  """T/hkv_hashtable_ops_test.py:76-100: a capacity that is no power of two, reserved_key_start_bit set."""
  torch, de = env
  t = de.get_variable("hkv_basic", key_dtype=torch.int64, value_dtype=torch.int32, initializer=0, dim=8, init_size=1024,
                      kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(max_capacity=99999,
                                                                                     reserved_key_start_bit=1)))
  assert int(t.size()) == 0
  t.clear()
  k = torch.tensor([-1, -2, -3, 0, 2**63 - 1, -2**63], device="cuda")   # incl. HKV's reserved key patterns
  t.upsert(k, torch.arange(6, dtype=torch.int32, device="cuda")[:, None].repeat(1, 8))
  assert int(t.size()) == 6
  got, ex = t.lookup(k, return_exists=True)
  assert bool(ex.all()) and got[:, 0].tolist() == [0, 1, 2, 3, 4, 5]


@pytest.mark.parametrize("vdtype", ["float32", "int32", "int64", "int8"])
def test_insert_sizes_all_dims(env, vdtype):
  """T/hkv_hashtable_ops_test.py:248-290: 18 upserts of 85 new keys each, size == (i+1)*85, for every dim."""
  torch, de = env
  dt = getattr(torch, vdtype)
  for dim in [1, 2, 4, 8, 10, 16, 32, 64, 100, 200]:
    t = de.get_variable("hkv_insert_%s_%d" % (vdtype, dim), key_dtype=torch.int64, value_dtype=dt,
                        initializer=np.array([-1]), dim=dim, init_size=102400,
                        kv_creator=de.HkvHashTableCreator(config=de.HkvHashTableConfig(init_capacity=102400,
                                                                                       max_capacity=102400)))
    for i in range(18):
      k = torch.arange(85 * i, 85 * (i + 1), device="cuda")
      v = (k % 100)[:, None].repeat(1, dim).to(dt)
      t.upsert(k, v)
      assert int(t.size()) == (i + 1) * 85
    k = torch.arange(0, 85 * 18 + 5, device="cuda")
    got, ex = t.lookup(k, return_exists=True)
    assert ex.tolist() == [True] * (85 * 18) + [False] * 5
    assert torch.equal(got[:85 * 18], (k[:85 * 18] % 100)[:, None].repeat(1, dim).to(dt))
    assert bool((got[85 * 18:] == -1).all())
    t.clear()
    assert int(t.size()) == 0
