"""Checkpoint paths the round-1 review found untested: loading KV files into a bounded Hkv table, and the
optimizer slots (co-located state vectors) surviving save -> load.  Reference: K/cuckoo_hashtable_op.cc:310-505 (file
format), K/hkv_hashtable_op_gpu.cu.cc:573-640 (GPU load clears first), PY/dynamic_embedding_optimizer.py:870-904 (slot
variables are tables of their own and are checkpointed like any other)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  import tfra_amd.dynamic_embedding as de
  assert torch.cuda.is_available()
  return torch, de


def test_hkv_save_load_round_trip_at_max_capacity(env, tmp_path):
  """A default HkvHashTable is bounded from creation (slots == max_capacity): load must not need duplicate-safe inserts."""
  torch, de = env
  dim, cap = 8, 30_000
  t = de.HkvHashTable(torch.int64, torch.float32, torch.full((dim,), -1.0), init_capacity=cap, max_capacity=cap, device="cuda:0",
                      dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="ckpt_hkv")
  keys = torch.arange(1, 14_001, dtype=torch.int64, device="cuda")   # load 0.47: nothing is evicted (T/hkv_hashtable_ops_test.py:572-625)
  vals = (keys[:, None] * torch.ones(dim, device="cuda")).float()
  t.insert(keys, vals)
  n = int(t.size().item())
  assert n == 14_000
  t.save_to_file_system(str(tmp_path), file_name="hkv_table", dirpath_env=None)
  assert os.path.getsize(tmp_path / "hkv_table-keys") == n * 8
  assert os.path.getsize(tmp_path / "hkv_table-values") == n * dim * 4
  t2 = de.HkvHashTable(torch.int64, torch.float32, torch.full((dim,), -1.0), init_capacity=cap, max_capacity=cap, device="cuda:0",
                       dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="ckpt_hkv2")
  t2.insert(torch.tensor([999_999], device="cuda"), torch.zeros((1, dim), device="cuda"))   # cleared by the load
  got = t2.load_from_file_system(str(tmp_path), file_name="hkv_table", dirpath_env=None)
  assert got == n == int(t2.size().item())
  k1, v1 = t.export()
  k2, v2 = t2.export()
  o1, o2 = torch.argsort(k1), torch.argsort(k2)
  assert torch.equal(k1[o1], k2[o2]) and torch.equal(v1[o1], v2[o2])
  assert t2._table.slot_census()["locked"] == 0


@pytest.mark.parametrize("shards_after", [1, 2])
def test_optimizer_slots_survive_save_and_load(env, tmp_path, shards_after):
  """Adam's m and v (fields 1, 2 of the rows) are written next to the embedding under the reference's slot-variable
  names and restored — also when the shard count changes between save and load."""
  torch, de = env
  dim = 8
  opt = de.optimizers.Adam(1e-2)
  deo = de.DynamicEmbeddingOptimizer(opt)
  var = de.Variable(dim=dim, name="emb/w", initializer=0.25, devices=["cuda:0"], **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  rng = np.random.default_rng(0)
  ids = torch.from_numpy(rng.integers(0, 500, size=4096)).cuda()
  for _ in range(3):
    deo.apply_sparse(var, ids, torch.from_numpy(rng.standard_normal((4096, dim)).astype(np.float32)).cuda())
  var.save_to_file_system(str(tmp_path), optimizer=deo)
  names = sorted(os.listdir(tmp_path))
  assert "emb_w_mht_1of1-keys" in names and "emb_w_Adam_m_mht_1of1-values" in names and "emb_w_Adam_v_mht_1of1-keys" in names
  uk = torch.unique(ids)
  want = [var.lookup(uk)] + [v.lookup(uk) for v in var.get_slot_variables(deo)]
  assert float(want[1].abs().max()) > 0 and float(want[2].abs().max()) > 0   # the slots really hold state
  var2 = de.Variable(dim=dim, name="emb/w", initializer=0.25, devices=["cuda:0"] * shards_after,
                     **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  var2.load_from_file_system(str(tmp_path), optimizer=deo)
  got = [var2.lookup(uk)] + [v.lookup(uk) for v in var2.get_slot_variables(deo)]
  assert int(var2.size().item()) == uk.numel()
  for a, b in zip(want, got):
    assert torch.equal(a, b)
  # training resumes from the restored state: the next step matches on both variables, bit for bit
  g = torch.from_numpy(rng.standard_normal((4096, dim)).astype(np.float32)).cuda()
  deo2 = de.DynamicEmbeddingOptimizer(de.optimizers.Adam(1e-2))
  deo2.iterations = deo.iterations
  deo.apply_sparse(var, ids, g)
  deo2.apply_sparse(var2, ids, g)
  assert torch.equal(var.lookup(uk), var2.lookup(uk))


def test_slot_file_fallback_matches_exact_names_only(env, tmp_path):
  """Restoring WITHOUT the optimizer object finds `<param>_<Opt>_<slot>` files by exact name and maps them to the row
  fields by slot NAME; sibling variables in the same directory (`emb_2`, `emb_user`: same prefix) are never taken for
  this variable's state, and a checkpoint that has no slot files next to such a sibling loads without an error."""
  torch, de = env
  dim = 4
  opt = de.optimizers.Adam(1e-2)
  deo = de.DynamicEmbeddingOptimizer(opt)
  kw = de.DynamicEmbeddingOptimizer.variable_kwargs(opt)
  var = de.Variable(dim=dim, name="emb", initializer=0.25, devices=["cuda:0"], **kw)
  sib = de.Variable(dim=dim, name="emb_2", initializer=0.5, devices=["cuda:0"])            # sorts before emb_Adam_m
  sib2 = de.Variable(dim=dim, name="emb_user", initializer=0.5, devices=["cuda:0"])
  ids = torch.arange(64, device="cuda")
  g = torch.ones((64, dim), device="cuda")
  for _ in range(2):
    deo.apply_sparse(var, ids, g)
  foreign = torch.arange(1000, 1100, device="cuda")
  sib.upsert(foreign, torch.full((100, dim), 7.0, device="cuda"))
  sib2.upsert(foreign, torch.full((100, dim), 9.0, device="cuda"))
  var.save_to_file_system(str(tmp_path), optimizer=deo)
  sib.save_to_file_system(str(tmp_path))
  sib2.save_to_file_system(str(tmp_path))
  want = [var.lookup(ids)] + [v.lookup(ids) for v in var.get_slot_variables(deo)]
  var2 = de.Variable(dim=dim, name="emb", initializer=0.25, devices=["cuda:0"], **kw)
  var2.load_from_file_system(str(tmp_path))                     # optimizer=None: the named files, by exact name
  assert int(var2.size().item()) == 64                           # none of the siblings' keys came along
  got = [var2.lookup(ids)] + [v.lookup(ids) for v in var2.get_slot_variables(deo)]
  for a, b in zip(want, got):
    assert torch.equal(a, b)
  # a checkpoint saved WITHOUT slot files, siblings present: loads, slots start at their initial values
  d2 = tmp_path / "noslots"
  d2.mkdir()
  plain = de.Variable(dim=dim, name="emb", initializer=0.25, devices=["cuda:0"])
  plain.upsert(ids, want[0])
  plain.save_to_file_system(str(d2))
  sib.save_to_file_system(str(d2))
  var3 = de.Variable(dim=dim, name="emb", initializer=0.25, devices=["cuda:0"], **kw)
  var3.load_from_file_system(str(d2), optimizer=deo)
  assert int(var3.size().item()) == 64 and torch.equal(var3.lookup(ids), want[0])
  # state files of another optimizer (FTRL's accum / linear) do not cover Adam's fields: an error, not a silent reset
  d3 = tmp_path / "other_opt"
  d3.mkdir()
  fo = de.optimizers.Ftrl(0.05)
  fdeo = de.DynamicEmbeddingOptimizer(fo)
  fvar = de.Variable(dim=dim, name="emb", initializer=0.25, devices=["cuda:0"], **de.DynamicEmbeddingOptimizer.variable_kwargs(fo))
  fdeo.apply_sparse(fvar, ids, g)
  fvar.save_to_file_system(str(d3), optimizer=fdeo)
  var4 = de.Variable(dim=dim, name="emb", initializer=0.25, devices=["cuda:0"], **kw)
  with pytest.raises(ValueError, match="optimizer-state files"):
    var4.load_from_file_system(str(d3), optimizer=deo)


def test_one_global_step_for_tables_sharing_an_optimizer(env):
  """begin_step(): N tables written with the parameters of ONE step advance `iterations` once (TF's apply_gradients)."""
  torch, de = env
  opt = de.optimizers.Adam(1e-2)
  deo = de.DynamicEmbeddingOptimizer(opt)
  tabs = [de.Variable(dim=4, name="shared_opt_%d" % i, initializer=0.0, devices=["cuda:0"],
                      **de.DynamicEmbeddingOptimizer.variable_kwargs(opt)) for i in range(3)]
  ids = torch.arange(16, device="cuda")
  g = torch.ones((16, 4), device="cuda")
  p = deo.begin_step()
  for v in tabs:
    deo.apply_sparse(v, ids, g, p)
  assert deo.iterations == 1
  one = de.Variable(dim=4, name="shared_opt_single", initializer=0.0, devices=["cuda:0"],
                    **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  deo1 = de.DynamicEmbeddingOptimizer(de.optimizers.Adam(1e-2))
  deo1.apply_sparse(one, ids, g)
  for v in tabs:
    assert torch.equal(v.lookup(ids), one.lookup(ids))
