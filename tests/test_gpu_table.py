"""GPU parity tests proper: the HIP table (called through the C ABI via tfra_amd) against the
oracle on the same seeded inputs — bit-exact for keys / exists / sizes / copied values and for
accum (one add per element, index order)."""
import numpy as np
import pytest

import oracle
from tests import kats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
  from tests import hip_adapter
  return hip_adapter


GPU_DIMS = [1, 2, 4, 8, 10, 16, 32, 64, 100, 128, 200]  # T/dynamic_embedding_variable_test.py:404


@pytest.mark.parametrize("dim", GPU_DIMS)
@pytest.mark.parametrize("dtype", [np.float32, np.int32, np.int64, np.int8, np.float16, np.float64])
def test_k1_k2(hip, dim, dtype):
  kats.kat_k1_upsert_remove_lookup_export(hip.hip_factory(), dim, dtype)
  kats.kat_k2_find_with_exists_and_accum(hip.hip_factory(), dim, dtype)


def test_k3_k4_k6_k11_k15(hip):
  f = hip.hip_factory()
  kats.kat_k3_vector_default(f)
  kats.kat_k4_export_insert_roundtrip(f)
  kats.kat_k6_shape_validation(f)
  kats.kat_k11_import_export_cardinality(f)
  kats.kat_k15_repeat_insert_idempotent(f)


def test_signature_mismatch_raises(hip):
  import torch
  t = hip.HipTable(4).t
  with pytest.raises(TypeError, match="Signature mismatch"):
    t.insert(torch.zeros(3, dtype=torch.int32, device="cuda:0"), torch.zeros(3, 4, device="cuda:0"))
  with pytest.raises(TypeError, match="Signature mismatch"):
    t.remove(torch.zeros(3, dtype=torch.int32, device="cuda:0"))


@pytest.mark.parametrize("dim,dtype,init", [(16, np.float32, 0), (64, np.float32, 64), (3, np.int32, 16),
                                            (130, np.float32, 0), (5, np.int8, 0), (7, np.float16, 0),
                                            (2, np.int64, 0), (33, np.float64, 0)])
def test_random_ops_match_oracle(hip, dim, dtype, init):
  """insert (with duplicates: last wins) / find / accum (with duplicates: index order) / remove
  sequences incl. the sentinel key values, growth from a tiny table, vs the C port."""
  rng = np.random.default_rng(99 + dim)
  a = oracle.CpuTable(dim, dtype, kind="port")
  b = hip.HipTable(dim, dtype, init_size=init)
  universe = rng.integers(-2**62, 2**62, size=5000, dtype=np.int64)
  i64 = np.iinfo(np.int64)
  universe[:6] = [0, -1, i64.min, i64.min + 1, i64.max, i64.min + 2]

  def vals(n):
    if np.issubdtype(np.dtype(dtype), np.floating):
      return (rng.standard_normal((n, dim)) * 4).astype(dtype)
    hi = 100 if np.dtype(dtype) == np.int8 else 10**6
    return rng.integers(-hi // 4, hi // 4, size=(n, dim)).astype(dtype)

  for step in range(48):
    n = int(rng.integers(1, 1500))
    keys = rng.choice(universe, size=n)
    op = step % 4
    if op == 0:
      v = vals(n)
      a.insert(keys, v); b.insert(keys, v)
    elif op == 1:
      d = vals(n)
      (va, ea), (vb, eb) = a.find(keys, d, True), b.find(keys, d, True)
      np.testing.assert_array_equal(ea, eb)
      np.testing.assert_array_equal(va.view(np.uint8), vb.view(np.uint8))
      np.testing.assert_array_equal(a.find(keys, d[0]).view(np.uint8), b.find(keys, d[0]).view(np.uint8))
    elif op == 2:
      v = vals(n)
      ex = rng.random(n) < 0.5
      a.accum(keys, v, ex); b.accum(keys, v, ex)
    else:
      a.remove(keys[: n // 3]); b.remove(keys[: n // 3])
    assert a.size() == b.size(), "step %d" % step
  (ka, va), (kb, vb) = a.export_sorted(), b.export_sorted()
  np.testing.assert_array_equal(ka, kb)
  np.testing.assert_array_equal(va.view(np.uint8), vb.view(np.uint8))


def test_bfloat16_copy_and_accum(hip):
  rng = np.random.default_rng(5)
  dim = 24
  a = oracle.CpuTable(dim, bf16=True)
  b = hip.HipTable(dim, bf16=True)
  keys = rng.integers(0, 400, size=300).astype(np.int64)
  bits = (rng.standard_normal((300, dim)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
  a.insert(keys, bits); b.insert(keys, bits)
  uk = np.unique(keys)
  delta = (rng.standard_normal((uk.size, dim)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
  ex = np.ones(uk.size, bool)
  a.accum(uk, delta, ex); b.accum(uk, delta, ex)
  (ka, va), (kb, vb) = a.export_sorted(), b.export_sorted()
  np.testing.assert_array_equal(ka, kb)
  np.testing.assert_array_equal(va, vb)


def test_reference_engine_agrees_at_scale(hip):
  """1M keys, dim 16 (config C1 shape): HIP vs the REAL reference engine (cuckoohash_map.hh)."""
  if not oracle.available("reference"):
    pytest.skip("oracle/_ref not built")
  rng = np.random.default_rng(7)
  n, dim = 1_000_000, 16
  keys = rng.permutation(np.arange(1, 4 * n, 4, dtype=np.int64))[:n] * 2654435761 % (2**61)
  keys = np.unique(keys)
  vals = rng.standard_normal((keys.size, dim)).astype(np.float32)
  a = oracle.CpuTable(dim, np.float32, kind="reference", init_size=keys.size, threads=8)
  b = hip.HipTable(dim, np.float32)
  a.insert(keys, vals); b.insert(keys, vals)
  assert a.size() == b.size() == keys.size
  q = np.concatenate([rng.choice(keys, 200_000), rng.integers(2**61, 2**62, 100_000)])
  d = np.full(dim, -1, np.float32)
  (va, ea), (vb, eb) = a.find(q, d, True), b.find(q, d, True)
  np.testing.assert_array_equal(ea, eb)
  np.testing.assert_array_equal(va, vb)
  (ka, xa), (kb, xb) = a.export_sorted(), b.export_sorted()
  np.testing.assert_array_equal(ka, kb)
  np.testing.assert_array_equal(xa, xb)


def test_golden_vectors(hip):
  """Replay the committed op sequences generated from the real reference engine."""
  import glob, os
  files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_ops_*.npz")))
  assert files, "golden fixtures missing"
  from tests.golden.replay import replay
  for f in files:
    replay(f, lambda dim, dtype: hip.HipTable(dim, dtype))


def test_export_batch_windows_and_idempotence(hip):
  """export over slot windows covers every live key exactly once; erase->reinsert is idempotent."""
  import torch
  rng = np.random.default_rng(3)
  t = hip.HipTable(8, np.float32)
  keys = np.unique(rng.integers(-2**40, 2**40, size=40000).astype(np.int64))
  vals = rng.standard_normal((keys.size, 8)).astype(np.float32)
  t.insert(keys, vals)
  tab = t.t._table
  cap = tab.capacity()
  got = []
  import ctypes
  from tfra_amd import _capi
  from tfra_amd.dynamic_embedding.table_ops import _ptr, _stream
  for off in range(0, cap, 9973):
    kbuf = torch.empty(9973, dtype=torch.int64, device="cuda:0")
    vbuf = torch.empty((9973, 8), dtype=torch.float32, device="cuda:0")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    _capi.call("tfra_table_export_batch", tab._h, min(9973, cap - off), off, _ptr(cnt), _ptr(kbuf), _ptr(vbuf), None,
               _stream(tab.device))
    c = int(cnt.item())
    got.append((kbuf[:c].cpu().numpy(), vbuf[:c].cpu().numpy()))
  gk = np.concatenate([g[0] for g in got]); gv = np.concatenate([g[1] for g in got])
  o = np.argsort(gk)
  np.testing.assert_array_equal(gk[o], keys)
  np.testing.assert_array_equal(gv[o], vals)
  t.remove(keys[::2]); assert t.size() == keys.size - keys[::2].size
  t.insert(keys[::2], vals[::2]); assert t.size() == keys.size
  ek, ev = t.export_sorted()
  np.testing.assert_array_equal(ek, keys); np.testing.assert_array_equal(ev, vals)


def test_save_load_files_match_reference_format(hip, tmp_path):
  """-keys / -values are raw native-endian arrays (K12 + cuckoo_hashtable_op.cc:310-505)."""
  rng = np.random.default_rng(11)
  dim = 5
  keys = np.unique(rng.integers(-10**9, 10**9, size=20000).astype(np.int64))
  vals = rng.standard_normal((keys.size, dim)).astype(np.float32)
  t = hip.HipTable(dim, np.float32)
  t.insert(keys, vals)
  n = t.t.save_to_file_system(str(tmp_path), file_name="tbl", dirpath_env=None, buffer_size=4096)
  assert n == keys.size
  fk = np.fromfile(tmp_path / "tbl-keys", dtype=np.int64)
  fv = np.fromfile(tmp_path / "tbl-values", dtype=np.float32).reshape(-1, dim)
  o = np.argsort(fk)
  np.testing.assert_array_equal(fk[o], keys); np.testing.assert_array_equal(fv[o], vals)
  # a file written the way the reference writes it loads into a fresh table
  perm = rng.permutation(keys.size)
  keys[perm].tofile(tmp_path / "ref-keys"); vals[perm].tofile(tmp_path / "ref-values")
  t2 = hip.HipTable(dim, np.float32)
  assert t2.t.load_from_file_system(str(tmp_path), file_name="ref", dirpath_env=None, buffer_size=3000) == keys.size
  ek, ev = t2.export_sorted()
  np.testing.assert_array_equal(ek, keys); np.testing.assert_array_equal(ev, vals)
  # load_entire_dir: both files into one table (K12)
  t3 = hip.HipTable(dim, np.float32)
  t3.t.load_from_file_system(str(tmp_path), file_name="x", dirpath_env=None, load_entire_dir=True)
  assert t3.size() == keys.size


def test_accum_hot_duplicates_on_device_index_order(hip):
  """accum with a key that repeats thousands of times in one call (Zipf head), mixed exists flags: applied in index
  order ON THE DEVICE (stable sort + per-key walk; no host copy of the keys) — bit-exact vs the sequential reference
  engine, including insert-then-add and add-before-insert (dropped) sequences.  cuckoohash_map.hh:619-633."""
  rng = np.random.default_rng(2024)
  dim = 24
  a = oracle.CpuTable(dim, np.float32, kind="reference" if oracle.available("reference") else "port")
  b = hip.HipTable(dim, np.float32, init_size=0)
  n = 40_000
  keys = (rng.zipf(1.2, size=n) % 3000).astype(np.int64) * 104729 - 7
  assert np.bincount((keys + 7) // 104729).max() > 3000
  for rnd in range(2):
    v = (rng.standard_normal((n, dim)) * 3).astype(np.float32)
    ex = rng.random(n) < (0.3 if rnd == 0 else 0.8)
    a.accum(keys, v, ex); b.accum(keys, v, ex)
    assert a.size() == b.size()
  (ka, va), (kb, vb) = a.export_sorted(), b.export_sorted()
  np.testing.assert_array_equal(ka, kb)
  np.testing.assert_array_equal(va.view(np.uint8), vb.view(np.uint8))
