"""CPU, gloo, world_size 2: the multi-process id-routing logic of AllToAllEmbedding
(partition -> alltoall ids -> local lookup -> alltoall rows -> un-permute; backward mirror) with
TEST DOUBLES for the two device pieces (the HIP table and the HIP front-end kernels cannot run
without a GPU): the local shard is the CPU oracle, the partition/gather/scatter ops are numpy
restatements.  Checks against the single-process model of the reference's
`__alltoall_embedding_lookup__` (oracle/frontends.py, PY/shadow_embedding_ops.py:397-447)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

DIM = 6


class _CpuOps:
  """numpy stand-ins with the signatures of tfra_amd.dynamic_embedding.device_ops."""

  def __init__(self, use_reduce=True):
    self.use_reduce = use_reduce

  def partition(self, ids, world, mode, n_dev=None):
    from oracle import frontends as ofe
    k = ids.numpy()
    n_all = k.size
    if n_dev is not None:
      k = k[:int(n_dev)]
    owner = ofe.default_partition_fn(k, world, gpu_mode=(mode == 0))
    perm = np.concatenate([np.nonzero(owner == r)[0] for r in range(world)]).astype(np.int32)
    counts = np.array([(owner == r).sum() for r in range(world)], dtype=np.int64)
    pad = np.zeros(n_all - k.size, dtype=np.int64)     # the device op returns length-n buffers
    return (torch.from_numpy(np.concatenate([k[perm], pad])), torch.from_numpy(np.concatenate([perm, pad.astype(np.int32)])),
            torch.from_numpy(counts))

  def unique_no_sync(self, ids):
    from oracle import frontends as ofe
    u, idx = ofe.unique(ids.numpy())
    buf = np.zeros(ids.numel(), dtype=np.int64)
    buf[:u.size] = u
    return torch.from_numpy(buf), torch.from_numpy(idx.astype(np.int32)), torch.tensor(u.size, dtype=torch.int64)

  def segment_sum(self, grads, idx, cnt, max_segments):
    out = np.zeros((max_segments, grads.shape[-1]), dtype=np.float32)
    g, ix = grads.numpy(), idx.numpy()
    for i in range(ix.size):                           # input order, one fp32 add each
      out[ix[i]] += g[i]
    return torch.from_numpy(out)

  def can_reduce_by_key(self, n, dim):
    return self.use_reduce

  def reduce_by_key(self, ids, grads):
    """Distinct ids in DESCENDING key order (any fixed order is allowed), sums in input order."""
    from oracle import optimizers as oopt
    u, g, _ = oopt.segment_sum_by_key(ids.numpy().reshape(-1), grads.numpy())
    order = np.argsort(-u, kind="stable")
    n = ids.numel()
    kb = np.zeros(n, dtype=np.int64); kb[:u.size] = u[order]
    gb = np.zeros((n, grads.shape[-1]), dtype=np.float32); gb[:u.size] = g[order]
    return torch.from_numpy(kb), torch.from_numpy(gb), torch.tensor(u.size, dtype=torch.int64)

  def gather_rows(self, rows, idx):
    return rows[idx.long()]

  def scatter_rows(self, rows, perm):
    out = torch.empty_like(rows)
    out[perm.long()] = rows
    return out


class _OracleShard:
  """de.Variable stand-in over the CPU oracle: lookup + SGD write-back with duplicate sums."""

  def __init__(self, dim):
    import oracle
    self.t = oracle.CpuTable(dim)
    self.dim = dim

  def lookup(self, ids):
    return torch.from_numpy(self.t.find(ids.numpy(), np.full(self.dim, -1.0, np.float32)))


class _SgdOpt:
  def __init__(self, lr):
    self.lr = lr

  def apply_sparse(self, shard, ids, grads, p=None):
    from oracle import optimizers as oopt
    uniq, g, _ = oopt.segment_sum_by_key(ids.numpy(), grads.numpy())
    p = shard.t.find(uniq, np.full(shard.dim, -1.0, np.float32))
    shard.t.insert(uniq, oopt.sgd(p, g, self.lr))


def _all_keys():
  return np.arange(0, 400, dtype=np.int64) * 7919 - 1000


def _ids(rank):
  rng = np.random.default_rng(100 + rank)
  return rng.choice(np.concatenate([_all_keys(), np.arange(10**6, 10**6 + 50)]), size=(5, 37 + 11 * rank))


def _grads(rank, n):
  return np.random.default_rng(200 + rank).standard_normal((n, DIM)).astype(np.float32)


def _worker(rank, world, port, q, dedup):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from oracle import frontends as ofe
  from tfra_amd.dynamic_embedding.distributed import AllToAllEmbedding
  shard = _OracleShard(DIM)
  keys = _all_keys()
  mine = keys[ofe.default_partition_fn(keys, world) == rank]
  shard.t.insert(mine, np.tile(mine[:, None].astype(np.float32) * 0.5, (1, DIM)))
  emb = AllToAllEmbedding(shard, partition_mode=0, ops=_CpuOps(use_reduce=(dedup != 'segsum')), dedup=bool(dedup))
  ids = torch.from_numpy(_ids(rank))
  out = emb.lookup(ids)
  g = _grads(rank, ids.numel())
  emb.apply_gradients(_SgdOpt(0.1), torch.from_numpy(g))
  dist.barrier()
  k, v = shard.t.export_sorted()
  q.put((rank, out.numpy(), k, v))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize("dedup", [True, "segsum", False])
def test_alltoall_lookup_and_write_back_world2(dedup):
  world, port = 2, 29511 + (os.getpid() % 200) + {True: 300, 'segsum': 600, False: 0}[dedup]
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, world, port, q, dedup)) for r in range(world)]
  for p in procs:
    p.start()
  res = {}
  for _ in range(world):
    r, out, k, v = q.get(timeout=120)
    res[r] = (out, k, v)
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  # single-process model
  import oracle
  from oracle import frontends as ofe
  from oracle import optimizers as oopt
  keys = _all_keys()
  owner = ofe.default_partition_fn(keys, world)
  tabs = [oracle.CpuTable(DIM) for _ in range(world)]
  for r in range(world):
    mine = keys[owner == r]
    tabs[r].insert(mine, np.tile(mine[:, None].astype(np.float32) * 0.5, (1, DIM)))
  ids = [_ids(r) for r in range(world)]
  exp = ofe.alltoall_lookup_model(tabs, ids, np.full(DIM, -1.0, np.float32))
  for r in range(world):
    np.testing.assert_array_equal(res[r][0].reshape(-1, DIM), exp[r])
    assert res[r][0].shape == ids[r].shape + (DIM,)
  # write-back: each owner receives the grads of every rank for its keys, rank order = alltoall order
  for owner_rank in range(world):
    ks, gs = [], []
    for src in range(world):
      flat = ids[src].reshape(-1)
      sel = ofe.default_partition_fn(flat, world) == owner_rank
      k_src, g_src = flat[sel], _grads(src, flat.size)[sel]
      if dedup:                                        # each source sums its repeats before routing
        k_src, g_src, _ = oopt.segment_sum_by_key(k_src, g_src)
      ks.append(k_src); gs.append(g_src)
    ks, gs = np.concatenate(ks), np.concatenate(gs)
    uniq, gsum, _ = oopt.segment_sum_by_key(ks, gs)
    p = tabs[owner_rank].find(uniq, np.full(DIM, -1.0, np.float32))
    tabs[owner_rank].insert(uniq, oopt.sgd(p, gsum, 0.1))
    ek, ev = tabs[owner_rank].export_sorted()
    np.testing.assert_array_equal(res[owner_rank][1], ek)
    np.testing.assert_allclose(res[owner_rank][2], ev, rtol=1e-6, atol=1e-6)


# ---- the routed assign step (csrc/tfra_aroute.hip) restated with numpy doubles over gloo ------------------------------------------
def _ra_batch(rank, step):
  rng = np.random.default_rng(40 * step + rank)
  n = 300 + 17 * rank + (step % 3) * 5
  ids = (rng.zipf(1.3, size=n).astype(np.int64) % 400) * 7919 - 1234
  ids[: n // 6] = 7919 * 3 - 1234                       # a hot id both ranks write every step
  rng.shuffle(ids)
  vals = (np.arange(n, dtype=np.float32) + 1000.0 * (step + 1) + 500.0 * rank)[:, None].repeat(DIM, 1)
  return ids, vals


def _ra_worker(rank, world, port, q, steps):
  """One rank of the route as the C driver issues it, the device pieces replaced by the numpy restatement (oracle/frontends.py
  route_plan) and the shard by the CPU oracle: per step gather(values at the last positions) -> alltoall(values) -> the owner writes
  the PREVIOUS batch back (source-major: the highest rank's row wins) and looks THIS batch's ids up -> alltoall(rows) -> gather."""
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  import oracle
  from oracle import frontends as ofe
  shard = oracle.CpuTable(DIM)
  dflt = np.zeros(DIM, np.float32)

  def a2a(send, counts_send, width, dtype):
    cs = torch.tensor(counts_send, dtype=torch.int64)
    cr = torch.empty(world, dtype=torch.int64)
    dist.all_to_all_single(cr, cs)
    recv = torch.empty((int(cr.sum()),) + ((width,) if width else ()), dtype=dtype)
    dist.all_to_all_single(recv, torch.from_numpy(np.ascontiguousarray(send)), cr.tolist(), cs.tolist())
    return recv.numpy(), cr.numpy()

  prev = None        # (last positions, send counts, values, received ids) of the batch looked up by the previous step
  looked = []
  for s in range(steps + 1):
    if prev is not None:                                  # write-back half of the step
      lastpos, counts, vals, recv_ids_prev = prev
      got, _ = a2a(vals[lastpos], counts.tolist(), DIM, torch.float32)
      shard.insert(recv_ids_prev, got)                    # source-major, sequential: the last one wins
    if s == steps:
      break
    ids, vals = _ra_batch(rank, s)
    owner_major, lastpos, pos2row, counts = ofe.route_plan(ids, world)
    recv_ids, rc = a2a(owner_major, counts.tolist(), 0, torch.int64)
    rows = shard.find(recv_ids, dflt)                     # the owner's lookup (after the write-back above: lookup i+1 sees update i)
    back, _ = a2a(rows, rc.tolist(), DIM, torch.float32)
    looked.append(back[pos2row])
    prev = (lastpos, counts, vals, recv_ids)
  k, v = shard.export_sorted()
  q.put((rank, looked, k, v))
  dist.barrier()
  dist.destroy_process_group()


def test_routed_assign_step_world2_equals_one_table():
  world, steps, port = 2, 5, 29811 + (os.getpid() % 150)
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_ra_worker, args=(r, world, port, q, steps)) for r in range(world)]
  for p in procs:
    p.start()
  res = {}
  for _ in range(world):
    r, looked, k, v = q.get(timeout=120)
    res[r] = (looked, k, v)
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  import oracle
  from oracle import frontends as ofe
  tab = oracle.CpuTable(DIM)
  dflt = np.zeros(DIM, np.float32)
  for s in range(steps):
    batches = [_ra_batch(r, s) for r in range(world)]
    rows = ofe.routed_assign_model(tab, [b[0] for b in batches], [b[1] for b in batches], dflt)
    for r in range(world):
      np.testing.assert_array_equal(res[r][0][s], rows[r], err_msg="rank %d step %d" % (r, s))
  ek, ev = tab.export_sorted()
  gk = np.concatenate([res[r][1] for r in range(world)])
  gv = np.concatenate([res[r][2] for r in range(world)])
  o = np.argsort(gk)
  np.testing.assert_array_equal(gk[o], ek)                # every key on exactly one shard
  np.testing.assert_array_equal(gv[o], ev)
  for r in range(world):
    assert np.all(ofe.default_partition_fn(res[r][1], world) == r)
