"""CPU: the C port and the real reference engine (cuckoohash_map.hh compiled in place) both
reproduce the reference's KATs, and agree with each other on seeded random op sequences."""
import numpy as np
import pytest

import oracle
from tests import kats

KINDS = ["port"] + (["reference"] if oracle.available("reference") or True else [])


def factory(kind):
  def make(dim, dtype=np.float32, **kw):
    return oracle.CpuTable(dim, dtype, kind=kind, **kw)
  return make


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("dim", [1, 8, 16, 128])
@pytest.mark.parametrize("dtype", [np.float32, np.int32, np.int64, np.int8, np.float64])
def test_k1_k2(kind, dim, dtype):
  kats.kat_k1_upsert_remove_lookup_export(factory(kind), dim, dtype)
  kats.kat_k2_find_with_exists_and_accum(factory(kind), dim, dtype)


@pytest.mark.parametrize("kind", KINDS)
def test_k3_k4_k6_k11_k15(kind):
  kats.kat_k3_vector_default(factory(kind))
  kats.kat_k4_export_insert_roundtrip(factory(kind))
  kats.kat_k6_shape_validation(factory(kind))
  kats.kat_k11_import_export_cardinality(factory(kind))
  kats.kat_k15_repeat_insert_idempotent(factory(kind), n=5000)


@pytest.mark.parametrize("dim,dtype", [(16, np.float32), (64, np.float32), (3, np.int32), (130, np.float32)])
def test_port_matches_reference_engine_random_ops(dim, dtype):
  """Differential: random insert / accum / remove / find sequences with duplicate keys."""
  rng = np.random.default_rng(1234 + dim)
  a = oracle.CpuTable(dim, dtype, kind="port")
  b = oracle.CpuTable(dim, dtype, kind="reference")
  universe = rng.integers(-2**62, 2**62, size=3000, dtype=np.int64)
  universe[:4] = [0, -1, np.iinfo(np.int64).min, np.iinfo(np.int64).max]
  for step in range(60):
    n = int(rng.integers(1, 700))
    keys = rng.choice(universe, size=n)
    op = step % 4
    if op == 0:
      vals = (rng.standard_normal((n, dim)) * 10).astype(dtype)
      a.insert(keys, vals); b.insert(keys, vals)
    elif op == 1:
      defaults = (rng.standard_normal((n, dim)) * 10).astype(dtype)
      (va, ea), (vb, eb) = a.find(keys, defaults, True), b.find(keys, defaults, True)
      np.testing.assert_array_equal(ea, eb)
      np.testing.assert_array_equal(va, vb)
      d1 = defaults[0]
      np.testing.assert_array_equal(a.find(keys, d1), b.find(keys, d1))
    elif op == 2:
      vod = (rng.standard_normal((n, dim)) * 10).astype(dtype)
      ex = rng.random(n) < 0.5
      a.accum(keys, vod, ex); b.accum(keys, vod, ex)
    else:
      a.remove(keys[: n // 3]); b.remove(keys[: n // 3])
    assert a.size() == b.size()
  (ka, va), (kb, vb) = a.export_sorted(), b.export_sorted()
  np.testing.assert_array_equal(ka, kb)
  np.testing.assert_array_equal(va, vb)
  # chunked dump covers the table exactly once (SaveToFileSystem pattern)
  got = []
  off = 0
  while True:
    k, _ = a.dump(off, 257)
    if k.size == 0:
      break
    got.append(k); off += k.size
  assert np.array_equal(np.sort(np.concatenate(got)), ka)


def test_port_replays_golden_vectors_from_reference_engine():
  """The committed fixtures were produced by the real reference engine; the port must replay them."""
  import glob, os
  from tests.golden.replay import replay
  files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_ops_*.npz")))
  assert len(files) >= 5
  for f in files:
    replay(f, lambda dim, dtype: oracle.CpuTable(dim, dtype, kind="port"))


# ---- size restriction restatement (T/restrict_policies_test.py) -------------------------------------
def test_restrict_oracle_timestamp_kat():
  """T/restrict_policies_test.py:166-229: ids 0..5 then 4..8 a second later; keep 5 -> 4..8 survive."""
  from oracle import frontends as ofe
  orc, table = ofe.RestrictPolicyOracle("timestamp"), {}
  for now, ids in ((100, np.arange(6)), (101, np.arange(4, 9))):
    orc.apply_update(ids, now=now)
    table.update({int(k): 1 for k in ids})
  assert orc.apply_restriction(table, 5, trigger=100) == [] and len(table) == 9
  orc.apply_restriction(table, 5, trigger=5)
  assert sorted(table) == [4, 5, 6, 7, 8] and sorted(orc.status) == [4, 5, 6, 7, 8]


def test_restrict_oracle_frequency_kat():
  """T/restrict_policies_test.py:232-327: counts after 0..2 then 1..3 are [1,2,2,1]; keep 2 of 0..8."""
  from oracle import frontends as ofe
  orc = ofe.RestrictPolicyOracle("frequency")
  orc.apply_update(np.arange(3)); orc.apply_update(np.arange(1, 4))
  assert [orc.status[k] for k in range(4)] == [1, 2, 2, 1]
  orc.apply_update(np.array([3, 3, 3]))
  assert orc.status[3] == 2                       # lookup/+1/insert: a repeated id counts once per call
  orc, table = ofe.RestrictPolicyOracle("frequency"), {}
  for ids in (np.arange(6), np.arange(4, 9)):
    orc.apply_update(ids)
    table.update({int(k): 1 for k in ids})
  orc.apply_restriction(table, 2, trigger=2)
  assert sorted(table) == [4, 5]


def test_restrict_select_is_stable_and_signed():
  from oracle import frontends as ofe
  keys = np.array([10, 11, 12, 13, 14], dtype=np.int64)
  st = np.array([3, -1, 3, -1, 0], dtype=np.int32)
  assert ofe.restrict_select(keys, st, 2).tolist() == [11, 13, 14]
  assert ofe.restrict_select(keys, st, 9).tolist() == []
  assert ofe.restrict_select(keys, st, 0).tolist() == [11, 13, 14, 10, 12]


# ---- DynamicPartition / DynamicStitch restatements vs the reference's own known answers ------------
def test_dynamic_partition_kats_oracle():
  from oracle import frontends as ofe
  from tests import kats_partition
  for name, data, parts, num, want in kats_partition.partition_cases():
    got = ofe.dynamic_partition(data, parts, num)
    assert len(got) == num, name
    for g, w in zip(got, want):
      w = np.asarray(w, dtype=np.float64).reshape(g.shape) if np.asarray(w).size == 0 else np.asarray(w)
      np.testing.assert_array_equal(g, w, err_msg=name)


def test_dynamic_stitch_kats_oracle():
  from oracle import frontends as ofe
  from tests import kats_partition
  for name, idx, data, want in kats_partition.stitch_cases():
    got = ofe.dynamic_stitch(idx, data)
    np.testing.assert_array_equal(got, np.asarray(want).reshape(got.shape), err_msg=name)


def test_safe_embedding_lookup_sparse_kats_oracle():
  """T/dynamic_embedding_ops_test.py:1007-1324 on the NumPy restatement (negative ids are legal keys)."""
  from oracle import frontends as ofe
  from tests import kats_sparse
  E = kats_sparse.embeddings(np.random.default_rng(0))
  t = oracle.CpuTable(kats_sparse.DIM)
  ks = np.array(sorted(E), dtype=np.int64)
  t.insert(ks, np.stack([E[int(k)] for k in ks]))
  zeros = np.zeros(kats_sparse.DIM, np.float32)
  kats_sparse.run(lambda idx, ids, shape, w, d: ofe.safe_embedding_lookup_sparse(t, idx, ids, shape, zeros, w, "mean", d), E)
