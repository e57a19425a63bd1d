"""Multi-GPU id routing: hash-sharded tables, one process per GPU, alltoall over RCCL/xGMI.

Mirror of `HvdAllToAllEmbedding` / `HvdVariable.__alltoall_embedding_lookup__`
(DE/python/keras/layers/embedding.py:545-594, PY/shadow_embedding_ops.py:365-447):

    forward : unique ids -> partition by owner rank -> alltoall(ids, splits) -> local table lookup
              -> alltoall(rows, splits=remote_sizes) -> un-permute -> expand to input order
    backward: sum grads of repeated ids -> permute owner-major -> alltoall(grads)
              -> local fused optimizer write-back (sums the <= world parts of a key)

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

  def partition(self, ids, world, mode, n_dev=None):
    return self._o.partition(ids, world, mode, n_dev=n_dev)

  def unique_no_sync(self, ids):
    return self._o.unique_no_sync(ids)

  def segment_sum(self, grads, idx, cnt, max_segments):
    return self._o.segment_sum(grads, idx, cnt, max_segments)

  def can_reduce_by_key(self, n, dim):
    return dim % 4 == 0 and dim <= 256 and n <= (1 << 18)

  def reduce_by_key(self, ids, grads):
    return self._o.reduce_by_key(ids, grads)

  def gather_rows(self, rows, idx):
    return self._o.gather_rows(rows, idx)

  def scatter_rows(self, rows, perm):
    return self._o.scatter_rows(rows, perm)


class AllToAllEmbedding:
  """One logical table sharded by key hash over the ranks of `group`.

  dedup=True (default): every rank sends each DISTINCT id of its batch once — `tf.unique` before the id
  alltoall, as the reference does (PY/shadow_embedding_ops.py:316 `embedding_lookup_unique`) — and sums
  the gradients of repeated ids locally before routing them back.  Under a Zipf head this bounds the
  skew to <= world copies of a hot key at its owner and cuts the alltoall payload by the batch's
  duplicate ratio (6x at Zipf-1.2, B = 131 072).  The unique count stays on the device; the ONLY host
  read per lookup is the pair of split-size vectors that alltoallv needs (as `hvd.alltoall(ids, splits)`).
  """

  def __init__(self, local, group=None, partition_mode=0, ops=None, dedup=True, force_collectives=False):
    self.local = local
    self.group = group
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    self.mode = partition_mode
    self.ops = ops if ops is not None else _DeviceOps()
    self.dedup = dedup
    # world == 1 normally short-circuits; `force_collectives` keeps the full route (used to exercise the
    # RCCL calls on a single-GPU box)
    self.passthrough = self.world == 1 and not (force_collectives and dist.is_initialized())
    self._route = None

  def _a2a(self, out, inp, out_splits=None, in_splits=None):
    """alltoall(v).  RCCL moves device buffers directly; the gloo backend (CPU tests, single-GPU
    smoke runs of the N>1 code path) has no device alltoall, so buffers are staged through the host."""
    if inp.is_cuda and dist.get_backend(self.group) == "gloo":
      o = torch.empty(out.shape, dtype=out.dtype)
      dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
      out.copy_(o)
    else:
      dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group)

  # -- id routing (PY/shadow_embedding_ops.py:397-422 __relocate_dense_feature__) ---------------
  def _exchange_counts(self, counts):
    recv = torch.empty_like(counts)
    self._a2a(recv, counts)
    return recv

  def route_ids(self, ids):
    """Returns the ids this rank must serve and remembers the routing for the way back."""
    ids = ids.reshape(-1)
    if self.passthrough:
      self._route = None
      return ids
    n = ids.numel()
    idx = cnt = None
    if self.dedup:
      uniq, idx, cnt = self.ops.unique_no_sync(ids)   # uniq: length-n buffer, first `cnt` entries valid
      owner_major, perm, counts = self.ops.partition(uniq, self.world, self.mode, n_dev=cnt)
    else:
      owner_major, perm, counts = self.ops.partition(ids, self.world, self.mode)
    recv_counts = self._exchange_counts(counts)
    # split sizes must be host-visible for alltoallv (as for hvd.alltoall(ids, splits))
    both = torch.stack([counts, recv_counts]).tolist()
    send, recv = [int(x) for x in both[0]], [int(x) for x in both[1]]
    u = sum(send)                                      # = number of distinct ids when dedup, else n
    remote_ids = torch.empty(sum(recv), dtype=ids.dtype, device=ids.device)
    self._a2a(remote_ids, owner_major[:u], recv, send)
    self._route = (perm[:u], send, recv, n, idx, cnt, u, ids)
    return remote_ids

  def return_rows(self, rows):
    """Rows for the ids served here -> rows in the asking ranks' input order."""
    if self._route is None:
      return rows
    perm, send, recv, n, idx, cnt, u, _ = self._route
    back = torch.empty((u, rows.shape[-1]), dtype=rows.dtype, device=rows.device)
    self._a2a(back, rows.contiguous(), send, recv)
    out = self.ops.scatter_rows(back, perm)            # owner-major -> order of the routed ids
    if idx is not None:
      out = self.ops.gather_rows(out, idx)             # distinct ids -> every position of the batch
    return out

  def route_grads(self, grads):
    """Gradients w.r.t. the rows of the last lookup -> (keys, gradient rows) at the owner ranks.

    dedup: the gradients of one id's repeats are summed here first.  The parallel reduction
    (`reduce_by_key`, cost independent of how often the hottest id repeats) returns the distinct ids in
    its own order, so the keys travel with the rows (8 B next to a dim*4-B row); the per-owner counts
    are those of the forward pass — same id set, same partition function — so no new size exchange."""
    if self._route is None:
      return self._served, grads
    perm, send, recv, n, idx, cnt, u, ids = self._route
    g = grads.reshape(n, -1)
    if idx is None:
      owner_major = self.ops.gather_rows(g, perm)
      remote = torch.empty((sum(recv), g.shape[-1]), dtype=g.dtype, device=g.device)
      self._a2a(remote, owner_major, recv, send)
      return self._served, remote
    if self.ops.can_reduce_by_key(n, g.shape[-1]):
      keys_u, gsum, cnt2 = self.ops.reduce_by_key(ids, g)
      owner_keys, perm2, _ = self.ops.partition(keys_u, self.world, self.mode, n_dev=cnt2)
      owner_major = self.ops.gather_rows(gsum, perm2[:u])
      remote_keys = torch.empty(sum(recv), dtype=ids.dtype, device=ids.device)
      self._a2a(remote_keys, owner_keys[:u], recv, send)
    else:  # sequential per-id sums in tf.unique order: same order as the forward route
      gsum = self.ops.segment_sum(g, idx, cnt, max(u, 1))[:u]
      owner_major = self.ops.gather_rows(gsum, perm)
      remote_keys = self._served
    remote = torch.empty((sum(recv), owner_major.shape[-1]), dtype=owner_major.dtype, device=grads.device)
    self._a2a(remote, owner_major, recv, send)
    return remote_keys, remote

  # -- the two halves of a training step ---------------------------------------------------------
  def lookup(self, ids):
    """PY/shadow_embedding_ops.py:424-447 __alltoall_embedding_lookup__"""
    shape = tuple(ids.shape)
    served = self.route_ids(ids)
    self._served = served
    rows = self.local.lookup(served)
    out = self.return_rows(rows.reshape(served.numel(), -1))
    return out.reshape(shape + (out.shape[-1],))

  def apply_gradients(self, optimizer, grads, p=None):
    """Backward of the alltoall (Horovod's registered gradient) + local sparse write-back: the owner
    sums what the ranks sent for one key (<= world parts when dedup) and applies one fused update.
    p: the step's parameters (`optimizer.begin_step()`) when several embeddings share the optimizer."""
    keys, g = self.route_grads(grads)
    optimizer.apply_sparse(self.local, keys, g, p)


class RoutedPrefetchStep:
  """The sharded training step with the id-only half of the route done AHEAD on a second stream — the multi-GPU
  counterpart of `PrefetchStep`.

  Everything the route needs before rows or gradients exist depends on the ids alone: the distinct ids of the batch and
  the position -> distinct-id map (`tfra_unique`), their owner-major order and per-owner counts (`tfra_partition`), the
  count exchange and the host read of the split sizes, the id alltoall itself, the de-duplication plans of the local
  batch (for the gradient sums) and of the ids received from the other ranks (for the owner's fused update).  `feed()`
  starts that for a batch as soon as its ids exist (keep two batches fed ahead: the split sizes are then copied to pinned
  memory one whole step before they are needed, and the host never waits for them).  What is left on the critical path:

      lookup : local find of the received ids -> alltoall(rows) -> ONE gather (position -> row of the returned block)
      apply  : per-key gradient sums written straight into owner-major order (tfra_plan_reduce_to) -> alltoall(grads)
               -> tfra_table_apply_planned at the owner (sums the <= world parts of a key, one fused update)

  i.e. two collectives and three kernels around the local step instead of five collectives, a host read and eight
  kernels.  The keys are not sent again with the gradients: the owner keeps the ids it received for the lookup, and the
  gradient rows arrive in the same order.  Every rank must call feed / lookup / apply in the same sequence (the
  collectives are issued in program order).  Reference: PY/shadow_embedding_ops.py:397-447 (routes at lookup time,
  synchronously, through Horovod).

      rs = RoutedPrefetchStep(var, deo); rs.feed(ids0); rs.feed(ids1)
      for i in ...:
        rows = rs.lookup()                    # [n, dim] rows of the oldest fed batch, in its id order
        ...                                   # forward / backward of the model
        rs.apply(grads)                       # gradients of those rows: route + fused update at the owners
        rs.feed(ids_{i+2})                    # ids must be complete in memory (the second stream does not wait for yours)
  """
  NSLOTS = 4

  def __init__(self, var, optimizer, group=None, partition_mode=0, force_collectives=False):
    from .optimizer import DynamicEmbeddingOptimizer
    from .table_ops import SparsePlan
    if var.shard_num != 1 or not DynamicEmbeddingOptimizer.can_plan(var, 1):
      raise ValueError("RoutedPrefetchStep needs a single-shard fp32 local Variable with dim % 4 == 0, dim <= 256")
    optimizer._check(var)
    self.var, self.deo = var, optimizer
    self.group, self.mode = group, partition_mode
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    self.collectives = dist.is_initialized() and (self.world > 1 or force_collectives)
    self.t = var.tables[0]
    self.table = self.t._table
    self.dev = self.table.device
    self.ops = _DeviceOps()
    self.side = torch.cuda.Stream(device=self.dev)
    self.slots = [dict(plan_local=SparsePlan(self.dev, var.dim), plan_remote=SparsePlan(self.dev, var.dim), state=0,
                       host_counts=torch.empty((2, self.world), dtype=torch.int64).pin_memory()) for _ in range(self.NSLOTS)]
    self.head = 0      # oldest fed batch = the one lookup / apply work on
    self.tail = 0      # next free slot
    self.fed = 0       # batches fed and not yet applied
    self.default = self.t._default_value.to(device=self.dev, dtype=torch.float32).contiguous()
    self._iota = None

  def _a2a(self, out, inp, out_splits=None, in_splits=None):
    if not self.collectives:
      out.copy_(inp)
      return
    if inp.is_cuda and dist.get_backend(self.group) == "gloo":   # host-staged (tests on one GPU)
      o = torch.empty(out.shape, dtype=out.dtype)
      dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
      out.copy_(o)
    else:
      dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group)

  def feed(self, ids, ids_ready=True):
    """First half of the id-only route of one batch, on the second stream: distinct ids, owner-major order, count
    exchange, split sizes on their way to pinned memory.  ids_ready=False: the ids are still being produced on the
    current stream (the second stream then waits for it)."""
    if self.fed >= self.NSLOTS - 1:
      raise RuntimeError("RoutedPrefetchStep: %d batches are fed ahead already" % self.fed)
    sl = self.slots[self.tail]
    ids = torch.as_tensor(ids, device=self.dev).reshape(-1).to(torch.int64).contiguous()
    if not ids_ready:
      self.side.wait_stream(torch.cuda.current_stream(self.dev))
    if sl.get("done") is not None:
      self.side.wait_event(sl["done"])    # the slot's plans were last read by the step that applied it (NSLOTS steps ago)
    with torch.cuda.stream(self.side):
      uniq, idx, cnt = self.ops.unique_no_sync(ids)
      owner_major, perm, counts = self.ops.partition(uniq, self.world, self.mode, n_dev=cnt)
      recv_counts = torch.empty_like(counts)
      self._a2a(recv_counts, counts)
      sl["host_counts"].copy_(torch.stack([counts, recv_counts]), non_blocking=True)
      ev = sl.get("counts_ev") or torch.cuda.Event()
      ev.record(self.side)
      sl["plan_local"].build(ids, sync=False)
    for tns in (ids, uniq, idx, owner_major, perm, counts, recv_counts):
      tns.record_stream(self.side)
    sl.update(ids=ids, n=ids.numel(), idx=idx, owner_major=owner_major, perm=perm, counts_ev=ev, state=1)
    self.tail = (self.tail + 1) % self.NSLOTS
    self.fed += 1

  def _finish(self, sl):
    """Second half, once the split sizes are on the host (no wait when the batch was fed a step earlier): the id
    alltoall, the position -> returned-row map, the plan of the ids this rank serves."""
    if sl["state"] != 1:
      return
    sl["counts_ev"].synchronize()
    hc = sl["host_counts"]
    send, recv = [int(x) for x in hc[0]], [int(x) for x in hc[1]]
    u = sum(send)
    with torch.cuda.stream(self.side):
      remote_ids = torch.empty(sum(recv), dtype=torch.int64, device=self.dev)
      self._a2a(remote_ids, sl["owner_major"][:u].contiguous(), recv, send)
      # position -> row of the owner-major block: inverse of perm (owner-major j holds distinct id perm[j]), through idx —
      # two row moves of 4-byte rows (a scatter and a gather) instead of four framework ops
      if self._iota is None or self._iota.numel() < max(u, 1):
        self._iota = torch.arange(max(u, 1, sl["n"]), dtype=torch.int32, device=self.dev).reshape(-1, 1)
      inv = self.ops.scatter_rows(self._iota[:max(u, 1)], sl["perm"][:max(u, 1)] if u else self._iota[:1, 0])
      pos2row = self.ops.gather_rows(inv, sl["idx"]).reshape(-1)
      sl["plan_remote"].build(remote_ids, sync=False)
      ready = sl.get("ready") or torch.cuda.Event()
      ready.record(self.side)
    for tns in (remote_ids, inv, pos2row):
      tns.record_stream(self.side)
    sl.update(u=u, send=send, recv=recv, remote_ids=remote_ids, pos2row=pos2row, ready=ready, state=2)

  def lookup(self):
    if self.fed == 0:
      raise RuntimeError("RoutedPrefetchStep.lookup: no batch fed")
    sl = self.slots[self.head]
    self._finish(sl)
    main = torch.cuda.current_stream(self.dev)
    main.wait_event(sl["ready"])
    rows = self.t.lookup(sl["remote_ids"]).reshape(sl["remote_ids"].numel(), self.var.dim)
    back = torch.empty((max(sl["u"], 1), self.var.dim), dtype=rows.dtype, device=self.dev)
    self._a2a(back[:sl["u"]], rows.contiguous(), sl["send"], sl["recv"])
    return self.ops.gather_rows(back, sl["pos2row"])

  def apply(self, grads, p=None):
    if self.fed == 0:
      raise RuntimeError("RoutedPrefetchStep.apply: no batch fed")
    sl = self.slots[self.head]
    self._finish(sl)
    if p is None:
      p = self.deo.begin_step()
    main = torch.cuda.current_stream(self.dev)
    main.wait_event(sl["ready"])
    g = grads.reshape(sl["n"], self.var.dim)
    if g.dtype != torch.float32 or not g.is_contiguous():
      g = g.to(torch.float32).contiguous()
    gsum = torch.empty((max(sl["u"], 1), self.var.dim), dtype=torch.float32, device=self.dev)
    sl["plan_local"].reduce_to(g, sl["pos2row"], gsum, sync=False)
    nr = sl["remote_ids"].numel()
    remote = torch.empty((max(nr, 1), self.var.dim), dtype=torch.float32, device=self.dev)
    self._a2a(remote[:nr], gsum[:sl["u"]], sl["recv"], sl["send"])
    if nr:
      if getattr(self.var, "restrict_policy", None) is not None:
        self.var.restrict_policy.apply_update(sl["remote_ids"])
      self.table.apply_planned(p, sl["plan_remote"], remote[:nr], self.default, sync=False)
    done = sl.get("done") or torch.cuda.Event()
    done.record(main)
    sl["done"] = done
    sl["state"] = 0
    self.head = (self.head + 1) % self.NSLOTS
    self.fed -= 1
    if self.fed:      # the next batch: its split sizes arrived during this step — finish its route now, off the critical path
      self._finish(self.slots[self.head])


def _loaded_librccl():
  """Path of the librccl the process already has mapped (torch's), so the driver's communicators and torch's come from
  one copy of the library."""
  try:
    with open("/proc/self/maps") as f:
      for line in f:
        if "librccl" in line:
          return line.split()[-1]
  except OSError:
    pass
  import os
  cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
  return cand if os.path.exists(cand) else "librccl.so"


_HIP = None


def _hip():
  global _HIP
  if _HIP is None:
    import ctypes
    h = ctypes.CDLL("libamdhip64.so")
    h.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    h.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    h.hipDeviceSynchronize.argtypes = []
    _HIP = h
  return _HIP


class _StagedTransport:
  """Host-staged alltoallv through a torch.distributed group (gloo): lets the tests run the C driver with two ranks
  that share ONE GPU, where RCCL cannot form a communicator.  Not a product path."""

  def __init__(self, group, device):
    import ctypes
    from .. import _capi
    self.group, self.device = group, device
    self.error = None
    hip = _hip()
    world = dist.get_world_size(group)

    def fn(ctx, channel, send, send_bytes, recv, recv_bytes, stream):
      try:
        sb = [int(send_bytes[i]) for i in range(world)]
        rb = [int(recv_bytes[i]) for i in range(world)]
        hip.hipStreamSynchronize(stream)
        src = torch.empty(max(sum(sb), 1), dtype=torch.uint8)
        if sum(sb):
          assert hip.hipMemcpy(src.data_ptr(), send, sum(sb), 2) == 0
        dst = torch.empty(max(sum(rb), 1), dtype=torch.uint8)
        dist.all_to_all_single(dst[:sum(rb)], src[:sum(sb)], rb, sb, group=group)
        if sum(rb):
          assert hip.hipMemcpy(recv, dst.data_ptr(), sum(rb), 1) == 0
        return 0
      except BaseException as e:   # never unwind through the C frames
        self.error = e
        return 4
    self._fn = _capi.ALLTOALLV_FN(fn)
    self.struct = _capi.Transport(None, dist.get_rank(group), world, self._fn)


def _open_transport(kind, group, dev, rank, world):
  """The alltoallv transport of the C route drivers: 'rccl' = the driver's own pair of RCCL communicators (grouped ncclSend / ncclRecv
  over xGMI; the unique ids travel through the host framework's broadcast), 'staged' = host-staged through the torch.distributed
  group (tests: two ranks sharing ONE GPU cannot form an RCCL communicator).  Returns (staged, rccl_struct, rccl_ranks)."""
  import ctypes
  from .. import _capi
  if kind == "staged":
    return _StagedTransport(group, dev), None, None
  if kind != "rccl":
    raise ValueError("transport: 'auto', 'rccl', 'staged' or None")
  lib = _loaded_librccl().encode()
  # rank 0 makes the two unique ids; a status byte travels with them so that a failure there raises on EVERY rank
  # instead of leaving the others waiting in the broadcast
  ids = torch.zeros(2 * _capi.RCCL_ID_BYTES + 1, dtype=torch.uint8)
  err0 = None
  if rank == 0:
    buf = (ctypes.c_char * (2 * _capi.RCCL_ID_BYTES))()
    try:
      for ch in range(2):
        _capi.call("tfra_rccl_unique_id", lib, ctypes.byref(buf, ch * _capi.RCCL_ID_BYTES))
      ids = torch.frombuffer(bytearray(buf.raw) + bytearray([1]), dtype=torch.uint8).clone()
    except Exception as e:   # noqa: BLE001 — reported below, on every rank
      err0 = e
  on_dev = dist.get_backend(group) == "nccl"
  ids_x = ids.to(dev) if on_dev else ids
  if world > 1:
    dist.broadcast(ids_x, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
  raw = bytes(ids_x.cpu().numpy().tobytes())
  if raw[-1] != 1:
    raise RuntimeError("route transport: rank 0 could not create the RCCL unique ids (%s)" % (err0 or "see rank 0"))
  raw = raw[:-1]
  rccl = _capi.Transport()
  _capi.call("tfra_rccl_transport_create", lib, raw, rank, world, dev.index or 0, ctypes.byref(rccl))
  n = ctypes.c_int(0)
  _capi.call("tfra_rccl_transport_ranks", ctypes.byref(rccl), ctypes.byref(n))
  if n.value != world:
    raise RuntimeError("route transport: the RCCL communicators span %d ranks, the process group %d" % (n.value, world))
  return None, rccl, n.value


class NativeRoutedStep:
  """`RoutedPrefetchStep` with the whole sequence issued from C (`tfra_route_*`, csrc/tfra_route.hip): three ctypes
  calls per step instead of ~18 plus four torch.distributed calls, and the collectives are grouped ncclSend/ncclRecv on
  the driver's own RCCL communicators (one for rows/gradients, one for the counts/ids of later batches).  Same
  results as `RoutedPrefetchStep` / `AllToAllEmbedding` (same kernels, same summation order).

      rs = NativeRoutedStep(var, deo); rs.feed(ids0); rs.feed(ids1); rs.feed(ids2)
      for i in ...:
        rows = rs.lookup(); ...; rs.apply(grads); rs.feed(ids_{i+3})

  Keep three batches fed ahead (a batch moves one stage of the route per step; fewer works, with stalls).  threaded: a
  helper thread launches the kernels of the id-only half (the collectives always come from the calling thread).

  transport: "rccl" (default when torch.distributed runs on nccl), "staged" (host-staged through the group; tests),
  None (single rank, device copies).  Reference: PY/shadow_embedding_ops.py:397-447."""

  def __init__(self, var, optimizer, group=None, partition_mode=0, force_collectives=False, transport="auto", max_batch=1 << 18,
               threaded=True, share_transport_of=None):
    import ctypes
    from .. import _capi
    from .optimizer import DynamicEmbeddingOptimizer
    if var.shard_num != 1 or not DynamicEmbeddingOptimizer.can_plan(var, 1):
      raise ValueError("NativeRoutedStep needs a single-shard fp32 local Variable with dim % 4 == 0, dim <= 256")
    optimizer._check(var)
    self.var, self.deo = var, optimizer
    self.t = var.tables[0]
    self.table = self.t._table
    self.dev = self.table.device
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    collectives = dist.is_initialized() and (self.world > 1 or force_collectives)
    if transport == "auto":
      transport = None if not collectives else ("rccl" if dist.get_backend(group) == "nccl" else "staged")
    self._staged = None
    self._rccl = None
    self._owns_transport = True
    tr = None
    self._transport_owner = None
    self.rccl_ranks = None   # ranks of the RCCL communicators as RCCL reports them (None: no RCCL transport)
    if share_transport_of is not None:
      # many tables, ONE pair of communicators (MultiTableRoutedStep): the collectives of every route are issued by the calling
      # thread in call order, the same on every rank, so routes may share a transport; only the first one owns (and destroys) it
      if threaded or getattr(share_transport_of, "_threaded", False):
        raise ValueError("NativeRoutedStep: routes that share a transport must not use helper threads (threaded=False on all of them): "
                         "one pair of communicators is only safe when every collective is issued by the calling thread")
      self._owns_transport = False
      self._transport_owner = share_transport_of   # keeps the owner (and its communicators) alive as long as this route is
      self._staged, self._rccl, self.rccl_ranks = share_transport_of._staged, share_transport_of._rccl, share_transport_of.rccl_ranks
      tr = ctypes.byref(self._rccl) if self._rccl is not None else (ctypes.byref(self._staged.struct) if self._staged is not None else None)
    elif transport in ("rccl", "staged"):
      self._staged, self._rccl, self.rccl_ranks = _open_transport(transport, group, self.dev, self.rank, self.world)
      tr = ctypes.byref(self._rccl) if self._rccl is not None else ctypes.byref(self._staged.struct)
    elif transport is not None:
      raise ValueError("transport: 'auto', 'rccl', 'staged' or None")
    self._threaded = bool(threaded)
    self._h = ctypes.c_void_p()
    _capi.call("tfra_route_create", self.table._h, tr, int(partition_mode), int(max_batch),
               0 if threaded else _capi.ROUTE_NO_THREAD, ctypes.byref(self._h))
    self.default = self.t._default_value.to(device=self.dev, dtype=torch.float32).contiguous()
    self._ids = []     # fed batches, oldest first (kept alive until applied)
    self._capi, self._ctypes = _capi, ctypes

  def _call(self, name, *args):
    try:
      self._capi.call(name, *args)
    except self._capi.TfraError:
      if self._staged is not None and self._staged.error is not None:
        e, self._staged.error = self._staged.error, None
        raise e
      raise

  def _stream(self):
    return self._ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(self.dev.index or 0))

  def feed(self, ids, ids_ready=True):
    ids = torch.as_tensor(ids, device=self.dev).reshape(-1).to(torch.int64).contiguous()
    self._call("tfra_route_feed", self._h, ids.numel(), self._ctypes.c_void_p(ids.data_ptr()), 1 if ids_ready else 0, self._stream())
    self._ids.append(ids)

  def served_ids(self):
    """The ids this rank serves for the oldest fed batch (a view of the driver's buffer, valid until its apply)."""
    p, n, u = self._ctypes.c_void_p(), self._ctypes.c_size_t(), self._ctypes.c_size_t()
    self._call("tfra_route_served_ids", self._h, self._ctypes.byref(p), self._ctypes.byref(n), self._ctypes.byref(u))
    return p.value, int(n.value), int(u.value)

  def lookup(self):
    if not self._ids:
      raise RuntimeError("NativeRoutedStep.lookup: no batch fed")
    n = self._ids[0].numel()
    out = torch.empty((n, self.var.dim), dtype=torch.float32, device=self.dev)
    self._call("tfra_route_lookup", self._h, self._ctypes.c_void_p(out.data_ptr()), self._ctypes.c_void_p(self.default.data_ptr()), self._stream())
    return out

  def apply(self, grads, p=None):
    if not self._ids:
      raise RuntimeError("NativeRoutedStep.apply: no batch fed")
    n = self._ids[0].numel()
    if p is None:
      p = self.deo.begin_step()
    g = grads.reshape(n, self.var.dim)
    if g.dtype != torch.float32 or not g.is_contiguous():
      g = g.to(torch.float32).contiguous()
    if getattr(self.var, "restrict_policy", None) is not None:
      ptr, nr, _ = self.served_ids()
      if nr:
        served = torch.empty(nr, dtype=torch.int64, device=self.dev)
        torch.cuda.current_stream(self.dev).synchronize()   # the ids arrived on the driver's second stream
        if _hip().hipMemcpy(served.data_ptr(), ptr, nr * 8, 3) != 0:
          raise RuntimeError("NativeRoutedStep: copy of the served ids failed")
        self.var.restrict_policy.apply_update(served)
    self._call("tfra_route_apply", self._h, self._ctypes.byref(p), self._ctypes.c_void_p(g.data_ptr()),
               self._ctypes.c_void_p(self.default.data_ptr()), self._stream())
    self._ids.pop(0)

  def close(self):
    if getattr(self, "_h", None) is not None and self._h:
      self._capi.call("tfra_route_destroy", self._h)
      self._h = None
    if self._rccl is not None and self._owns_transport:
      self._capi.call("tfra_rccl_transport_destroy", self._ctypes.byref(self._rccl))
    self._rccl = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


class MultiTableRoutedStep:
  """BASELINE configs[4] across GPUs: many embedding tables (a DLRM's 26), EACH hash-sharded over the ranks by the reference's
  default partitioner, one `tfra_route` per table on ONE shared transport (one pair of RCCL communicators for all tables), driven by
  single multi-table calls:

      ms = MultiTableRoutedStep(variables, deo); ms.feed(ids_per_table) x 3
      for ...: rows = ms.lookup(); ...; ms.apply(grads_per_table); ms.feed(next_ids_per_table)

  Per table the sequence and the results are `NativeRoutedStep`'s (reference: the per-table `__alltoall_embedding_lookup__` of
  PY/shadow_embedding_ops.py:397-447 + the optimizer's sparse apply); every rank calls with the tables in the same order, so the
  collectives of all tables interleave identically everywhere.  One optimizer step (`begin_step`) covers all tables of an apply."""

  def __init__(self, variables, optimizer, group=None, partition_mode=0, force_collectives=False, transport="auto", max_batch=1 << 18,
               threaded=False):
    self.deo = optimizer
    self.steps = []
    for v in variables:
      self.steps.append(NativeRoutedStep(v, optimizer, group=group, partition_mode=partition_mode, force_collectives=force_collectives,
                                         transport=transport, max_batch=max_batch, threaded=threaded,
                                         share_transport_of=self.steps[0] if self.steps else None))
    self.world = self.steps[0].world
    self.rccl_ranks = self.steps[0].rccl_ranks

  def feed(self, ids_list):
    for s_, ids in zip(self.steps, ids_list):
      s_.feed(ids)

  def lookup(self):
    return [s_.lookup() for s_ in self.steps]

  def apply(self, grads_list):
    p = self.deo.begin_step()
    for s_, g in zip(self.steps, grads_list):
      s_.apply(g, p)

  def close(self):
    for s_ in reversed(self.steps):   # the owner of the transport last
      s_.close()
    self.steps = []


class RoutedAssignStep:
  """The metric's step — lookup(B ids) + insert_or_assign(B ids, B rows), repeats: the last occurrence wins — on a table that is
  hash-sharded over the ranks (one process per GPU, this rank's shard = `table`), driven from C (`tfra_assign_route_*`,
  csrc/tfra_aroute.hip).  Same call pattern as `OverlapAssignStep`, with the batches announced ahead:

      rs = RoutedAssignStep(table); for k in range(5): rs.feed(ids[k])
      rows_0 = rs.step()                       # lookup(batch 0)
      for i in 1, 2, ...:
        rs.feed(ids[i + 4])
        rows_i = rs.step(values_{i-1})         # write-back of batch i-1 (the owner sees it before lookup i), lookup(batch i)
      rs.flush(values_last)

  Route and results: the reference's `__alltoall_embedding_lookup__` (PY/shadow_embedding_ops.py:397-447) for the lookup and
  `Variable.upsert`'s partitioning of keys AND values by owner (PY/dynamic_embedding_variable.py:772-800) for the write-back —
  one table that sees, per step, every rank's lookup and then rank 0's, rank 1's, ... insert_or_assign.  Per batch TWO launches (the route plan) for all
  id-only work (de-duplication, owner grouping, last positions, position map), ahead of the step; per step on the critical path:
  gather -> alltoall(values) -> the owner's overlapped step launch -> alltoall(rows) -> gather.

  With ONE rank and no forced collectives the route is the identity (as `dynamic_partition` with one shard is in the reference):
  the calls go straight to `tfra_table_step_overlap` (identity=True; the fed batches are its look-ahead).
  transport: "auto" | "rccl" | "staged" (tests) | "local" (one rank THROUGH the route driver, device copies where the alltoalls
  would be: what the route itself costs)."""

  def __init__(self, table, group=None, partition_mode=0, transport="auto", max_batch=1 << 18):
    import ctypes
    from .. import _capi
    self.t = table
    self.table = table._table if hasattr(table, "_table") else table
    self.dev = self.table.device
    self.dim, self.vdt = self.table.dim, self.table.value_dtype
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    if transport == "auto":
      transport = None if self.world == 1 else ("rccl" if dist.get_backend(group) == "nccl" else "staged")
    self.identity = transport is None
    self._staged = self._rccl = None
    self.rccl_ranks = None
    self._capi, self._ctypes = _capi, ctypes
    self._fed = []        # announced batches not yet looked up, oldest first
    self._pending = None  # (ids, values) of the batch looked up last (kept alive until written back)
    self._keep = None
    self.default = self.table._default_value
    if self.default.dtype != self.vdt or not self.default.is_contiguous():
      self.default = self.default.to(self.vdt).contiguous()
    self._h = None
    if self.identity:
      if self.world != 1:
        raise ValueError("RoutedAssignStep: transport=None needs a single rank")
      from .optimizer import OverlapAssignStep
      self._ovl = OverlapAssignStep(table)
      return
    tr = None
    if transport in ("rccl", "staged"):
      self._staged, self._rccl, self.rccl_ranks = _open_transport(transport, group, self.dev, self.rank, self.world)
      tr = ctypes.byref(self._rccl) if self._rccl is not None else ctypes.byref(self._staged.struct)
    elif transport != "local":
      raise ValueError("transport: 'auto', 'rccl', 'staged', 'local' or None")
    elif self.world != 1:
      raise ValueError("RoutedAssignStep: transport='local' needs a single rank")
    self._h = ctypes.c_void_p()
    _capi.call("tfra_assign_route_create", self.table._h, tr, int(partition_mode), int(max_batch), ctypes.byref(self._h))

  def _call(self, name, *args):
    try:
      self._capi.call(name, *args)
    except self._capi.TfraError:
      if self._staged is not None and self._staged.error is not None:
        e, self._staged.error = self._staged.error, None
        raise e
      raise

  def _stream(self):
    return self._ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(self.dev.index or 0))

  def _as_ids(self, ids):
    if torch.is_tensor(ids) and ids.dtype == torch.int64 and ids.dim() == 1 and ids.is_contiguous() and ids.device == self.dev:
      return ids
    return torch.as_tensor(ids, device=self.dev).reshape(-1).to(torch.int64).contiguous()

  def feed(self, ids, ids_ready=False):
    """ids_ready=False (default): the ids may still be in flight on the current stream — the driver's own stream waits for them (one
    event); True: they are complete (a host synchronisation lies between their producer and this call)."""
    ids = self._as_ids(ids)
    if not self.identity:
      self._call("tfra_assign_route_feed", self._h, ids.numel(), self._ctypes.c_void_p(ids.data_ptr()), 1 if ids_ready else 0, self._stream())
    self._fed.append(ids)

  def _values(self, values, n):
    if values.dtype != self.vdt or not values.is_contiguous() or values.shape != (n, self.dim):
      values = values.reshape(n, self.dim).to(self.vdt).contiguous()
    return values

  def step(self, values_prev=None, out=None):
    """rows of the oldest announced batch; values_prev [n_prev, dim] = what the batch of the PREVIOUS step writes back."""
    if not self._fed:
      raise RuntimeError("RoutedAssignStep.step: no batch fed")
    if self._pending is not None and values_prev is None:
      raise ValueError("RoutedAssignStep.step: the previous step's batch has not been written back: values_prev is None")
    ids = self._fed[0]
    n = ids.numel()
    vp = self._values(values_prev, self._pending[0].numel()) if self._pending is not None else None
    if self.identity:
      rows = self._identity_step(vp, out)
    else:
      rows = out if out is not None else torch.empty((n, self.dim), dtype=self.vdt, device=self.dev)
      self._call("tfra_assign_route_step", self._h, self._ctypes.c_void_p(rows.data_ptr()), self._ctypes.c_void_p(self.default.data_ptr()),
                 self._ctypes.c_void_p(vp.data_ptr()) if vp is not None else None, self._stream())
    self._keep = self._pending
    self._pending = (ids, vp)
    self._fed.pop(0)
    return rows

  def _identity_step(self, vp, out):
    # one rank, no route: tfra_table_step_overlap itself — the announced batches are its look-ahead (the values of a batch arrive one
    # call later, as in the routed form: the driver defers the write-back)
    from .table_ops import _stream
    o, c = self._ovl, self._ctypes
    ids = self._fed[0]
    n = ids.numel()
    rows = out if out is not None else torch.empty((n, self.dim), dtype=self.vdt, device=self.dev)
    nxt = self._fed[1] if len(self._fed) > 1 else None
    nx2 = self._fed[2] if len(self._fed) > 2 else None
    self._capi.check(o._fn(o._h, n, c.c_void_p(ids.data_ptr()), c.c_void_p(rows.data_ptr()), None, o._default_p, 0,
                           c.c_void_p(vp.data_ptr()) if vp is not None else None, None,
                           0 if nxt is None else nxt.numel(), c.c_void_p(nxt.data_ptr()) if nxt is not None else None,
                           0 if nx2 is None else nx2.numel(), c.c_void_p(nx2.data_ptr()) if nx2 is not None else None, _stream(self.dev)))
    return rows

  def flush(self, values_prev):
    if self._pending is None:
      return
    vp = self._values(values_prev, self._pending[0].numel())
    if self.identity:
      from .table_ops import _stream
      self._capi.call("tfra_table_step_overlap_flush", self._ovl._h, self._ctypes.c_void_p(vp.data_ptr()), None, _stream(self.dev))
    else:
      self._call("tfra_assign_route_flush", self._h, self._ctypes.c_void_p(vp.data_ptr()), self._stream())
    self._keep = (self._pending, vp)
    self._pending = None

  def stats(self):
    if self.identity:
      st = self._ovl.stats()
      return {"steps": st["overlapped"] + st["sequential"], "stalls": 0, "owner_overlapped": st["overlapped"], "owner_sequential": st["sequential"]}
    buf = (self._ctypes.c_uint64 * 6)()
    self._call("tfra_assign_route_stats", self._h, buf)
    return {"steps": buf[0], "stalls": buf[1], "owner_overlapped": buf[2], "owner_sequential": buf[3], "distinct_ids_last_batch": buf[4],
            "served_ids_last_batch": buf[5]}

  def close(self):
    if getattr(self, "_h", None):
      self._capi.call("tfra_assign_route_destroy", self._h)
      self._h = None
    if self._rccl is not None:
      self._capi.call("tfra_rccl_transport_destroy", self._ctypes.byref(self._rccl))
      self._rccl = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
