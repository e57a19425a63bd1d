"""Known-answer tests transcribed from the reference's own Python tests (SURVEY.md
appendix A).  Each KAT is a function taking a *table factory* ``make(dim, dtype, default)``
returning an object with the reference table-op surface (find/insert/accum/remove/size/
export_sorted); the same functions run against the C port, the real reference engine and
the HIP table, so the parity tests read like the reference's tests.

T = /root/reference/tensorflow_recommenders_addons/dynamic_embedding/python/kernel_tests
"""
import numpy as np


def kat_k1_upsert_remove_lookup_export(make, dim, dtype):
  """T/dynamic_embedding_variable_test.py:425-466"""
  t = make(dim, dtype)
  default = np.full(dim, -1, dtype)
  keys = np.array([0, 1, 2, 3], np.int64)
  vals = np.array([[k] * dim for k in range(4)], dtype)
  t.insert(keys, vals)
  assert t.size() == 4
  t.remove(np.array([1, 5], np.int64))
  assert t.size() == 3
  out = t.find(np.array([0, 1, 5], np.int64), default)
  np.testing.assert_array_equal(out, np.array([[0] * dim, [-1] * dim, [-1] * dim], dtype))
  ek, ev = t.export_sorted()
  np.testing.assert_array_equal(ek, [0, 2, 3])
  np.testing.assert_array_equal(ev, np.array([[0] * dim, [2] * dim, [3] * dim], dtype))


def kat_k2_find_with_exists_and_accum(make, dim, dtype):
  """T/dynamic_embedding_variable_test.py:500-561 (all four accumrase cases)"""
  t = make(dim, dtype)
  default = np.full(dim, -1, dtype)
  t.insert(np.array([0, 1, 2, 3], np.int64), np.array([[k] * dim for k in range(4)], dtype))
  _, ex = t.find(np.array([0, 1, 100, 3], np.int64), default, return_exists=True)
  np.testing.assert_array_equal(ex, [True, True, False, True])
  # "another process" races in between
  t.insert(np.array([100], np.int64), np.array([[99] * dim], dtype))
  t.remove(np.array([1], np.int64))
  old = np.array([[0] * dim, [1] * dim, [2] * dim, [3] * dim], dtype)
  new = np.array([[10] * dim, [11] * dim, [100] * dim, [13] * dim], dtype)
  # Variable.accum builds where(exists, new-old, new)  (PY/dynamic_embedding_variable.py:829-837)
  vod = np.where(ex[:, None], new - old, new).astype(dtype)
  t.accum(np.array([0, 1, 100, 3], np.int64), vod, ex)
  assert t.size() == 4
  ek, ev = t.export_sorted()
  np.testing.assert_array_equal(ek, [0, 2, 3, 100])
  np.testing.assert_array_equal(ev, np.array([[10] * dim, [2] * dim, [13] * dim, [99] * dim], dtype))


def kat_k3_vector_default(make):
  """T/dynamic_embedding_variable_test.py:1460-1496"""
  t = make(2, np.int32)
  default = np.array([-1, -2], np.int32)
  t.insert(np.array([0, 1, 2, 3], np.int64), np.array([[0, 1], [2, 3], [4, 5], [6, 7]], np.int32))
  assert t.size() == 4
  t.remove(np.array([3, 4], np.int64))
  assert t.size() == 3
  out = t.find(np.array([0, 1, 4], np.int64), default)
  np.testing.assert_array_equal(out, [[0, 1], [2, 3], [-1, -2]])
  ek, ev = t.export_sorted()
  np.testing.assert_array_equal(ek, [0, 1, 2])
  np.testing.assert_array_equal(ev, [[0, 1], [2, 3], [4, 5]])


def kat_k4_export_insert_roundtrip(make):
  """T/dynamic_embedding_variable_test.py:1498-1534"""
  default = np.array([-1, -1], np.int32)
  t1 = make(2, np.int32)
  t1.insert(np.array([0, 1, 2], np.int64), np.array([[0, 1], [2, 3], [4, 5]], np.int32))
  q = np.array([0, 1, 3], np.int64)
  np.testing.assert_array_equal(t1.find(q, default), [[0, 1], [2, 3], [-1, -1]])
  k, v = t1.export()
  t2 = make(2, np.int32)
  t2.insert(k, v)
  np.testing.assert_array_equal(t2.find(q, default), [[0, 1], [2, 3], [-1, -1]])


def kat_k6_shape_validation(make):
  """T/dynamic_embedding_variable_test.py:1536-1570: wrong value shapes raise."""
  import pytest
  t = make(2, np.int32)
  keys = np.array([0, 1, 2], np.int64)
  for bad in ([0, 1, 2, 3, 4, 5], [[0, 1, 2], [3, 4, 5]], [[0, 1], [2, 3]], [[0], [2], [4]]):
    with pytest.raises(ValueError, match="Expected shape"):
      t.insert(keys, np.array(bad, np.int32))


def kat_k11_import_export_cardinality(make):
  """T/cuckoo_hashtable_ops_test.py:76-99: 168 keys, dim 3, int32."""
  t = make(3, np.int32)
  keys = np.arange(168, dtype=np.int64) * 7 - 300
  vals = np.arange(168 * 3, dtype=np.int32).reshape(168, 3)
  t.import_values(keys, vals)
  assert t.size() == 168
  ek, ev = t.export_sorted()
  np.testing.assert_array_equal(ek, np.sort(keys))
  np.testing.assert_array_equal(ev, vals[np.argsort(keys)])


def kat_k15_repeat_insert_idempotent(make, dim=8, dtype=np.float32, n=50000):
  """T/hkv_hashtable_ops_test.py:572-625"""
  t = make(dim, dtype)
  keys = np.arange(n, dtype=np.int64)
  vals = np.tile((keys % 97).astype(dtype)[:, None], (1, dim))
  t.insert(keys, vals)
  assert t.size() == n
  t.insert(keys, vals)
  assert t.size() == n
