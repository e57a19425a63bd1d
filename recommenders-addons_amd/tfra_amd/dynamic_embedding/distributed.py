"""Multi-GPU id routing: hash-sharded tables, one process per GPU, alltoall over RCCL/xGMI.

Mirror of `HvdAllToAllEmbedding` / `HvdVariable.__alltoall_embedding_lookup__`
(DE/python/keras/layers/embedding.py:545-594, PY/shadow_embedding_ops.py:365-447):

    forward : partition ids by owner rank -> alltoall(ids, splits) -> local table lookup
              -> alltoall(rows, splits=remote_sizes) -> un-permute to input order
    backward: permute grads owner-major -> alltoall(grads) -> local fused optimizer write-back

The reference does this with `hvd.alltoall` from Python; here it is `torch.distributed`
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests) with the partition / un-permute being
device kernels (tfra_partition / tfra_gather_rows / tfra_scatter_rows).  Every key belongs to
exactly one shard and shards never talk during find/insert, so the only collectives are the
two (three with counts) alltoalls; dense all-reduce stays with the host framework.

`local` is the rank-local shard (a `de.Variable`); `ops` are the device front-end ops.  Both are
injectable so the routing logic can be exercised on CPU with stand-ins (tests/test_distributed.py).
"""
import torch
import torch.distributed as dist


class _DeviceOps:
  """Default: the HIP front-end kernels."""

  def __init__(self):
    from . import device_ops
    self._o = device_ops

  def partition(self, ids, world, mode):
    return self._o.partition(ids, world, mode)

  def gather_rows(self, rows, idx):
    return self._o.gather_rows(rows, idx)

  def scatter_rows(self, rows, perm):
    return self._o.scatter_rows(rows, perm)


class AllToAllEmbedding:
  """One logical table sharded by key hash over the ranks of `group`."""

  def __init__(self, local, group=None, partition_mode=0, ops=None):
    self.local = local
    self.group = group
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    self.mode = partition_mode
    self.ops = ops if ops is not None else _DeviceOps()
    self._route = None

  def _a2a(self, out, inp, out_splits=None, in_splits=None):
    """alltoall(v).  RCCL moves device buffers directly; the gloo backend (CPU tests, single-GPU
    smoke runs of the N>1 code path) has no device alltoall, so buffers are staged through the host."""
    if inp.is_cuda and dist.get_backend(self.group) == "gloo":
      o = torch.empty(out.shape, dtype=out.dtype)
      dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
      out.copy_(o)
    else:
      dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

  # -- id routing (PY/shadow_embedding_ops.py:397-422 __relocate_dense_feature__) ---------------
  def _exchange_counts(self, counts):
    recv = torch.empty_like(counts)
    self._a2a(recv, counts)
    return recv

  def route_ids(self, ids):
    """Returns the ids this rank must serve and remembers the routing for the way back."""
    ids = ids.reshape(-1)
    if self.world == 1:
      self._route = None
      return ids
    owner_major, perm, counts = self.ops.partition(ids, self.world, self.mode)
    recv_counts = self._exchange_counts(counts)
    # split sizes must be host-visible for alltoallv (as for hvd.alltoall(ids, splits))
    both = torch.stack([counts, recv_counts]).tolist()
    send, recv = [int(x) for x in both[0]], [int(x) for x in both[1]]
    remote_ids = torch.empty(sum(recv), dtype=ids.dtype, device=ids.device)
    self._a2a(remote_ids, owner_major, recv, send)
    self._route = (perm, send, recv, ids.numel())
    return remote_ids

  def return_rows(self, rows):
    """Rows for the ids served here -> rows in the asking ranks' input order."""
    if self._route is None:
      return rows
    perm, send, recv, n = self._route
    back = torch.empty((sum(send), rows.shape[-1]), dtype=rows.dtype, device=rows.device)
    self._a2a(back, rows.contiguous(), send, recv)
    return self.ops.scatter_rows(back, perm)

  def route_grads(self, grads):
    """Gradients w.r.t. the rows of the last lookup -> owner ranks (order of `route_ids` output)."""
    if self._route is None:
      return grads
    perm, send, recv, n = self._route
    owner_major = self.ops.gather_rows(grads.reshape(n, -1), perm)
    remote = torch.empty((sum(recv), owner_major.shape[-1]), dtype=grads.dtype, device=grads.device)
    self._a2a(remote, owner_major, recv, send)
    return remote

  # -- the two halves of a training step ---------------------------------------------------------
  def lookup(self, ids):
    """PY/shadow_embedding_ops.py:424-447 __alltoall_embedding_lookup__"""
    shape = tuple(ids.shape)
    served = self.route_ids(ids)
    self._served = served
    rows = self.local.lookup(served)
    out = self.return_rows(rows.reshape(served.numel(), -1))
    return out.reshape(shape + (out.shape[-1],))

  def apply_gradients(self, optimizer, grads):
    """Backward of the alltoall (Horovod's registered gradient) + local sparse write-back."""
    g = self.route_grads(grads)
    optimizer.apply_sparse(self.local, self._served, g)
