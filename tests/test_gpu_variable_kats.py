"""Small Variable-level cases of the reference's own test file that the K-list does not name
(T/dynamic_embedding_variable_test.py:1689-1830, 1380-1420, 1414-1440 of dynamic_embedding_ops_test.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  import tfra_amd.dynamic_embedding as de
  return torch, de


def test_three_independent_variables(env):
  """test_dynamic_embedding_variables :1689-1724"""
  torch, de = env
  keys = torch.tensor([0, 1, 2], device="cuda")
  values = torch.tensor([[0], [1], [2]], dtype=torch.int32, device="cuda")
  tabs = [de.get_variable("t19%d" % i, torch.int64, torch.int32, initializer=-1) for i in (1, 2, 3)]
  for t in tabs:
    t.upsert(keys, values)
  for t in tabs:
    assert int(t.size()) == 3
    assert t.lookup(torch.tensor([0, 1, 3], device="cuda")).tolist() == [[0], [1], [-1]]


def test_tensor_default_and_int_float(env):
  """test_dynamic_embedding_variable_with_tensor_default :1726-1744, ..._int_float :1790-1809"""
  torch, de = env
  t = de.get_variable("t200", torch.int64, torch.int32, initializer=torch.tensor(-1, dtype=torch.int32))
  t.upsert(torch.tensor([0, 1, 2], device="cuda"), torch.tensor([[0], [1], [2]], dtype=torch.int32, device="cuda"))
  assert int(t.size()) == 3
  assert t.lookup(torch.tensor([0, 1, 3], device="cuda")).tolist() == [[0], [1], [-1]]
  f = de.get_variable("t220", torch.int64, torch.float32, initializer=-1.0)
  assert int(f.size()) == 0
  f.upsert(torch.tensor([3, 7, 0], device="cuda"), torch.tensor([[7.5], [-1.2], [9.9]], device="cuda"))
  assert int(f.size()) == 3
  np.testing.assert_allclose(f.lookup(torch.tensor([7, 0, 11], device="cuda")).cpu().numpy(), [[-1.2], [9.9], [-1.0]])


def test_signature_mismatch_at_variable_level(env):
  """test_signature_mismatch :1746-1788: wrong key / value dtypes are rejected and leave the table untouched."""
  torch, de = env
  t = de.get_variable("t210", torch.int64, torch.int32, initializer=-1)
  keys = torch.tensor([0, 1, 2], device="cuda")
  values = torch.tensor([[0], [1], [2]], dtype=torch.int32, device="cuda")
  with pytest.raises((ValueError, TypeError)):
    t.upsert(torch.tensor([4.0, 5.0, 6.0], device="cuda"), values)
  with pytest.raises((ValueError, TypeError)):
    t.upsert(keys, torch.tensor([[0.5], [1.5], [2.5]], device="cuda"))
  assert int(t.size()) == 0
  t.upsert(keys, values)
  assert int(t.size()) == 3
  t.upsert(torch.tensor([0], device="cuda"), torch.tensor([[-1]], dtype=torch.int32, device="cuda"))   # scalar-like upsert
  assert int(t.size()) == 3
  assert t.lookup(torch.tensor([0], device="cuda")).tolist() == [[-1]]
  with pytest.raises((ValueError, TypeError)):
    t.lookup(torch.tensor([1, 2, 3], dtype=torch.int32, device="cuda"))


def test_random_initializer_for_misses(env):
  """test_dynamic_embedding_variable_with_random_init :1811-1829: a miss takes a fresh initializer row."""
  torch, de = env
  t = de.get_variable("t230", torch.int64, torch.float32, initializer=lambda shape: torch.rand(shape) + 5.0)
  t.upsert(torch.tensor([0, 1, 2], device="cuda"), torch.tensor([[0.0], [1.0], [2.0]], device="cuda"))
  assert int(t.size()) == 3
  r = t.lookup(torch.tensor([0, 1, 3], device="cuda")).cpu().numpy()
  assert r[0, 0] == 0.0 and r[1, 0] == 1.0 and 5.0 <= r[2, 0] < 6.0
  assert int(t.size()) == 3            # lookup never inserts


def test_get_variable_reuse_and_clear(env):
  """test_get_variable :1380-1405 (same name -> same object), test_dynamic_embedding_variable_clear
  (T/dynamic_embedding_ops_test.py:1414-1439)."""
  torch, de = env
  a = de.get_variable("t_reuse", torch.int64, torch.float32, dim=2, initializer=-1.0)
  b = de.get_variable("t_reuse", torch.int64, torch.float32, dim=2, initializer=-1.0)
  c = de.get_variable("t_reuse_other", torch.int64, torch.float32, dim=2, initializer=-1.0)
  assert a is b and a is not c
  a.upsert(torch.tensor([1, 2, 3], device="cuda"), torch.ones((3, 2), device="cuda"))
  assert int(b.size()) == 3
  a.clear()
  assert int(a.size()) == 0
  assert a.lookup(torch.tensor([1], device="cuda")).tolist() == [[-1.0, -1.0]]
