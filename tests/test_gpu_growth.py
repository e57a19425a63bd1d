"""Growth in place (tables whose storage is a mapped virtual address range: csrc/tfra_table.hip, Table::grow_in_place):
every bucket splits into its children where it is.  The table must come out exactly as the copying growth leaves it —
same keys, rows, scores, optimizer slots, size — for cuckoo and Hkv flavours, with side rows (sentinel keys) and erased
slots, over several doublings; and a table of more than a third of the HBM must grow where a second copy cannot exist.
Reference behaviour: libcuckoo's rehash (K/lib/cuckoo/cuckoohash_map.hh) and HKV's `reserve` keep every entry."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def mapped(monkeypatch):
  """Every table created inside the test lives in a mapped address range (threshold 0 MB)."""
  monkeypatch.setenv("TFRA_VMM_THRESHOLD_MB", "0")
  import torch
  import tfra_amd.dynamic_embedding as de
  return torch, de


def row_of(torch, keys, dim, dtype=None):
  j = torch.arange(dim, device=keys.device, dtype=torch.int64)
  return (((keys[:, None] * 2654435761 + j[None, :] * 40503) % 65521).to(torch.float32) / 65521.0 - 0.5).to(dtype or torch.float32)


@pytest.mark.parametrize("dim,dtype_name", [(8, "float32"), (33, "float16"), (64, "float32"), (3, "int64")])
def test_cuckoo_grows_in_place_and_keeps_everything(mapped, dim, dtype_name):
  torch, de = mapped
  dt = getattr(torch, dtype_name)
  t = de.CuckooHashTable(torch.int64, dt, torch.zeros(dim, dtype=dt), init_size=1024, device="cuda:0", dim=dim, name="grow_%s" % dtype_name)
  assert t._table.growth_stats()["mapped_range"] == 1
  rng = np.random.default_rng(dim)
  all_keys = np.unique(rng.integers(-2**62, 2**62, size=400_000).astype(np.int64))
  all_keys = np.concatenate([all_keys, np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).min + 1], np.int64)])   # the sentinel keys: side rows
  rng.shuffle(all_keys)
  live = {}
  pos = 0
  for step, n in enumerate([500, 3000, 20_000, 60_000, 100_000, 150_000]):
    k = torch.from_numpy(all_keys[pos:pos + n]).cuda()
    pos += n
    t.insert(k, row_of(torch, k, dim, dt) if dt.is_floating_point else (k[:, None] + torch.arange(dim, device="cuda")[None, :]))
    if step == 2:   # holes: erased slots must not come back
      gone = torch.from_numpy(all_keys[100:2100]).cuda()
      t.remove(gone)
  st = t._table.growth_stats()
  assert st["in_place"] >= 5 and st["in_place"] == st["growths"], st
  inserted = all_keys[:pos]
  gone = set(all_keys[100:2100].tolist())
  want = np.array([x for x in inserted.tolist() if x not in gone], np.int64)
  assert int(t.size().item()) == want.size
  ek, ev = t.export()
  ek = ek.cpu().numpy()
  assert np.array_equal(np.sort(ek), np.sort(want))
  kk = torch.from_numpy(want).cuda()
  got, ex = t.lookup(kk, return_exists=True)
  assert bool(ex.all())
  exp = row_of(torch, kk, dim, dt) if dt.is_floating_point else (kk[:, None] + torch.arange(dim, device="cuda")[None, :])
  assert torch.equal(got, exp)
  _, ex2 = t.lookup(torch.from_numpy(np.array(sorted(gone), np.int64)).cuda(), return_exists=True)
  assert not bool(ex2.any())
  assert t._table.check_errors() is None


def test_in_place_equals_copying_growth_with_optimizer_slots(mapped, monkeypatch):
  """The same training run on a table that grows in place and on one that grows by copying: identical tables (rows and
  Adam slots bit for bit) — growth moves rows, it does not touch them."""
  torch, de = mapped
  rng = np.random.default_rng(5)

  def run(name):
    opt = de.optimizers.Adam(1e-2)
    var = de.Variable(dim=16, name=name, initializer=0.125, init_size=512, devices=["cuda:0"], **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    deo = de.DynamicEmbeddingOptimizer(opt)
    r = np.random.default_rng(9)
    for step in range(12):
      ids = torch.from_numpy((r.zipf(1.1, size=20_000) % (40_000 * (step + 1))).astype(np.int64) * 7919 - 11).cuda()
      g = torch.from_numpy((r.standard_normal((20_000, 16)) * 0.01).astype(np.float32)).cuda()
      deo.apply_sparse(var, ids, g)
    return var

  a = run("grow_inplace")
  sa = a.tables[0]._table.growth_stats()
  monkeypatch.setenv("TFRA_VMM_THRESHOLD_MB", "-1")
  b = run("grow_copy")
  sb = b.tables[0]._table.growth_stats()
  assert sa["in_place"] >= 2 and sb["in_place"] == 0 and sb["growths"] >= 2 and sb["mapped_range"] == 0, (sa, sb)
  ka, va = a.tables[0]._table.export_all()[:2]
  kb, vb = b.tables[0]._table.export_all()[:2]
  ia, ib = torch.argsort(ka), torch.argsort(kb)
  assert ka.numel() > 50_000 and torch.equal(ka[ia], kb[ib]) and torch.equal(va[ia], vb[ib])
  for f in (1, 2):   # Adam's m and v
    fa = a.tables[0]._table.find(ka[ia], torch.zeros(16, device="cuda"), field=f)
    fb = b.tables[0]._table.find(kb[ib], torch.zeros(16, device="cuda"), field=f)
    assert torch.equal(fa, fb) and float(fa.abs().max()) > 0


def test_hkv_scores_survive_in_place_growth(mapped):
  torch, de = mapped
  dim = 4
  t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=2048, max_capacity=2048 * 64, device="cuda:0", dim=dim,
                      evict_strategy=de.HkvEvictStrategy.CUSTOMIZED, name="grow_hkv")
  rng = np.random.default_rng(1)
  keys = np.unique(rng.integers(1, 2**60, size=60_000).astype(np.int64))[:50_000]
  scores = rng.integers(1, 2**40, size=keys.size).astype(np.uint64)
  for lo in range(0, keys.size, 5000):
    k = torch.from_numpy(keys[lo:lo + 5000]).cuda()
    t._table.upsert(k, row_of(torch, k, dim), scores=torch.from_numpy(scores[lo:lo + 5000].astype(np.int64)).cuda(), unique_keys=True)
  st = t._table.growth_stats()
  assert st["in_place"] >= 2 and st["in_place"] == st["growths"], st
  assert int(t.size().item()) == keys.size       # capacity 131072 slots: nothing evicted
  ek, ev, es = t.export_with_scores(1 << 20)
  o = torch.argsort(ek)
  assert np.array_equal(ek[o].cpu().numpy(), keys)
  assert np.array_equal(es[o].cpu().numpy().astype(np.uint64), scores)
  assert torch.equal(ev[o], row_of(torch, ek[o], dim))


def test_table_past_a_third_of_hbm_grows_in_place():
  """100 GB of buckets -> 200 GB: old + new (300 GB) cannot coexist on a 288 GB device, the split in place can.  Rows are a
  closed-form function of the key; a sample of 16 M keys is checked after the growth."""
  import torch
  import tfra_amd.dynamic_embedding as de
  dim = 64
  bucket = 128 + 15 * dim * 4
  nb = int(100e9 // bucket)
  slots = nb * 15
  t = de.CuckooHashTable(torch.int64, torch.float32, torch.zeros(dim), init_size=int(slots * 0.74), device="cuda:0", dim=dim, name="grow_big")
  cap0 = t._table.capacity()
  assert t._table.growth_stats()["mapped_range"] == 1 and abs(cap0 - slots) < slots * 0.02
  n_total = int(cap0 * 0.95)       # past the hard threshold (92 % of the slots): the table has to double on the way
  chunk = 8_000_000
  g = torch.Generator(device="cuda"); g.manual_seed(3)
  done = 0
  while done < n_total:
    n = min(chunk, n_total - done)
    k = torch.arange(done, done + n, device="cuda", dtype=torch.int64) * 2654435761 + 17      # distinct (odd multiplier mod 2^64)
    t._table.upsert(k, row_of(torch, k, dim), unique_keys=True)
    done += n
  st = t._table.growth_stats()
  assert st["in_place"] == 1 and st["growths"] == 1, st
  assert t._table.capacity() >= 2 * cap0 - 4
  assert int(t.size().item()) == n_total
  assert t._table.check_errors() is None
  idx = torch.randint(0, n_total, (16_000_000,), device="cuda", generator=g)
  k = idx * 2654435761 + 17
  for lo in range(0, k.numel(), 4_000_000):
    kk = k[lo:lo + 4_000_000]
    got, ex = t.lookup(kk, return_exists=True)
    assert bool(ex.all()) and torch.equal(got, row_of(torch, kk, dim))
  _, ex = t.lookup(torch.arange(1, 100_000, device="cuda", dtype=torch.int64) * 2654435761 + 18, return_exists=True)
  assert int(ex.sum()) == 0


@pytest.mark.parametrize("threshold_mb", ["0", "-1"])          # growth in place / by copying
@pytest.mark.parametrize("init,maxcap", [(1000, 10_000), (2048, 2048 * 8), (5000, 7000), (100, 100_000)])
def test_bounded_table_grows_to_max_capacity_then_evicts(monkeypatch, init, maxcap, threshold_mb):
  """An Hkv table created below max_capacity: it grows until it can double no more (slots <= max_capacity and more than half
  of it), then evicts by score — the batch with the highest scores survives (T/hkv_hashtable_evict_test.py:527-573)."""
  monkeypatch.setenv("TFRA_VMM_THRESHOLD_MB", threshold_mb)
  import torch
  import tfra_amd.dynamic_embedding as de
  dim = 4
  t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=init, max_capacity=maxcap, device="cuda:0", dim=dim,
                      evict_strategy=de.HkvEvictStrategy.CUSTOMIZED, name="bounded_grow_%d_%d_%s" % (init, maxcap, threshold_mb))
  assert t._table.capacity() - 2 <= maxcap
  rng = np.random.default_rng(init)
  keys = np.unique(rng.integers(1, 2**60, size=4 * maxcap + 1000).astype(np.int64))[:3 * maxcap]
  rng.shuffle(keys)
  vip = keys[:maxcap // 4]                       # inserted first with the highest scores
  batches = [(vip, 1_000_000)] + [(keys[lo:lo + 2000], 10) for lo in range(vip.size, keys.size, 2000)]
  for k, score in batches:
    kk = torch.from_numpy(k).cuda()
    t._table.upsert(kk, row_of(torch, kk, dim), scores=torch.full((k.size,), score, dtype=torch.int64, device="cuda"), unique_keys=True)
    assert int(t.size().item()) <= maxcap
  slots = t._table.capacity() - 2
  assert maxcap // 2 < slots <= maxcap, (slots, maxcap)
  st = t._table.growth_stats()
  if init * 4 < maxcap:
    assert st["growths"] >= 1 and (st["in_place"] == st["growths"]) == (threshold_mb == "0"), st
  size = int(t.size().item())
  assert size > 0.85 * slots                       # ran full: the rest was evicted, nothing was refused
  assert t._table.check_errors() is None
  kv = torch.from_numpy(vip).cuda()
  got, ex = t.lookup(kv, return_exists=True)
  assert bool(ex.all()) and torch.equal(got, row_of(torch, kv, dim))
