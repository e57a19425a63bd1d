"""Sparse optimizers for `de.Variable` — the write-back half of the hot path.

Reference: `DynamicEmbeddingOptimizer` patches a stock TF optimizer so that, per step and per
embedding variable, it runs (1+S) table finds -> the stock dense apply kernel on the local
[U,dim] buffers -> (1+S) table upserts, the S slots living in S extra hash tables
(PY/dynamic_embedding_optimizer.py:134-242, create_slots :870-958).  Here the S slot vectors are
co-located with the embedding row (`Variable(aux_fields=S)`) and the whole sequence is ONE HIP
kernel per shard (`tfra_table_apply_optimizer`), preceded by the duplicate-id reduction the
reference gets from `_resource_apply_sparse_duplicate_indices` (:177-190): gradients of
duplicate ids are summed, then one update per key.  Update rules = TF's
ResourceApply{GradientDescent,Adam,Adagrad[V2],Ftrl}.  Slots of unseen keys start from the
slot initial value (zeros / `initial_accumulator_value`), state is per key (lazy), the step
counter for Adam's bias correction is global (PY/...optimizer.py:870-931).
"""
import ctypes
import math

import torch

from .. import _capi
from . import device_ops
from .variable import TrainableWrapper, Variable


class _Opt:
  slots = ()
  kind = None

  def aux_init(self):
    return (0.0, 0.0, 0.0, 0.0)

  def params(self, step):
    raise NotImplementedError


class SGD(_Opt):
  """tf.train.GradientDescentOptimizer / keras SGD (no momentum)."""
  kind = _capi.OPT_SGD
  slots = ()

  def __init__(self, learning_rate=0.01):
    self.lr = learning_rate

  def params(self, step):
    p = _capi.OptParams()
    p.kind, p.lr = self.kind, self.lr
    return p


class Adam(_Opt):
  """tf.train.AdamOptimizer (epsilon-hat form): slots m, v."""
  kind = _capi.OPT_ADAM
  slots = ("m", "v")

  def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
    self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon

  _CHUNK = 1024

  def _lr_t(self, step):
    """lr_t = lr*sqrt(1-b2^t)/(1-b1^t), evaluated in fp32 like TF's kernel prologue.  Computed for 1024 steps
    at a time (the same fp32 ufuncs element by element) so that the per-step host cost is a list lookup."""
    import numpy as np
    base = (step // self._CHUNK) * self._CHUNK
    cache = getattr(self, "_lr_cache", None)
    if cache is None or cache[0] != (base, self.lr, self.b1, self.b2):
      f32 = np.float32
      t = np.arange(base, base + self._CHUNK, dtype=np.float32)
      b1p = np.power(f32(self.b1), t).astype(np.float32)
      b2p = np.power(f32(self.b2), t).astype(np.float32)
      with np.errstate(divide="ignore", invalid="ignore"):
        lr_t = (f32(self.lr) * np.sqrt(f32(1) - b2p) / (f32(1) - b1p)).astype(np.float32)
      cache = ((base, self.lr, self.b1, self.b2), lr_t.tolist())
      self._lr_cache = cache
    return cache[1][step - base]

  def params(self, step):
    p = _capi.OptParams()
    p.kind, p.lr, p.beta1, p.beta2, p.eps = self.kind, self._lr_t(step), self.b1, self.b2, self.eps
    return p


class Adagrad(_Opt):
  """tf.train.AdagradOptimizer (epsilon=None) or keras Adagrad (epsilon=1e-7): slot accumulator."""
  kind = _capi.OPT_ADAGRAD
  slots = ("accumulator",)

  def __init__(self, learning_rate=0.001, initial_accumulator_value=0.1, epsilon=None):
    self.lr, self.init_acc, self.eps = learning_rate, initial_accumulator_value, epsilon

  def aux_init(self):
    return (self.init_acc, 0.0, 0.0, 0.0)

  def params(self, step):
    p = _capi.OptParams()
    p.kind, p.lr, p.eps = self.kind, self.lr, (-1.0 if self.eps is None else self.eps)
    return p


class Ftrl(_Opt):
  """tf.train.FtrlOptimizer: slots accum, linear."""
  kind = _capi.OPT_FTRL
  slots = ("accum", "linear")

  def __init__(self, learning_rate=0.001, learning_rate_power=-0.5, initial_accumulator_value=0.1,
               l1_regularization_strength=0.0, l2_regularization_strength=0.0):
    self.lr, self.lr_power, self.init_acc = learning_rate, learning_rate_power, initial_accumulator_value
    self.l1, self.l2 = l1_regularization_strength, l2_regularization_strength

  def aux_init(self):
    return (self.init_acc, 0.0, 0.0, 0.0)

  def params(self, step):
    p = _capi.OptParams()
    p.kind, p.lr, p.l1, p.l2, p.lr_power = self.kind, self.lr, self.l1, self.l2, self.lr_power
    return p


class Generic(_Opt):
  """Any dense update rule, run as the reference runs EVERY optimizer (PY/dynamic_embedding_optimizer.py:165-204):
  sum the gradients of repeated ids, (1+S) table finds, the rule on the dense [U, dim] tensors — host-framework
  math, as TF's ResourceApply* kernels are for the reference — and (1+S) table upserts.  Subclass and implement
  `update(step, p, g, *slots) -> (p_new, *slots_new)` (fp32 torch tensors), or pass `update=`.  The four rules
  with a fused HIP kernel (SGD, Adam, Adagrad, Ftrl) do not take this path."""
  kind = None

  def __init__(self, slots=(), slot_init=(), update=None):
    self.slots = tuple(slots)
    self._slot_init = tuple(float(x) for x in slot_init) + (0.0,) * (4 - len(slot_init))
    if len(self.slots) > 4:
      raise ValueError("at most 4 slot vectors per row")
    if update is not None:
      self.update = update

  def aux_init(self):
    return self._slot_init[:4]

  def params(self, step):
    return None      # no fused kernel: DynamicEmbeddingOptimizer runs `update` on dense tensors

  def update(self, step, p, g, *slots):
    raise NotImplementedError


class Momentum(Generic):
  """tf.train.MomentumOptimizer (ResourceApplyMomentum / Keras SGD with momentum): accum = momentum*accum + g;
  p -= lr*accum, or with use_nesterov: p -= lr*g + lr*momentum*accum."""

  def __init__(self, learning_rate=0.01, momentum=0.9, use_nesterov=False):
    super().__init__(slots=("momentum",), slot_init=(0.0,))
    self.lr, self.momentum, self.nesterov = learning_rate, momentum, use_nesterov

  def update(self, step, p, g, accum):
    accum = accum * self.momentum + g
    if self.nesterov:
      p = p - (g * self.lr + accum * (self.momentum * self.lr))
    else:
      p = p - accum * self.lr
    return p, accum


class RMSProp(Generic):
  """tf.train.RMSPropOptimizer (ResourceApplyRMSProp, non-centered): ms = rho*ms + (1-rho)*g^2;
  mom = momentum*mom + lr*g/sqrt(ms + eps); p -= mom.  `rms` starts at 1 (TF1) — pass initial_rms=0 for Keras."""

  def __init__(self, learning_rate=0.001, decay=0.9, momentum=0.0, epsilon=1e-10, initial_rms=1.0):
    super().__init__(slots=("rms", "momentum"), slot_init=(initial_rms, 0.0))
    self.lr, self.rho, self.momentum, self.eps = learning_rate, decay, momentum, epsilon

  def update(self, step, p, g, ms, mom):
    ms = ms + (g * g - ms) * (1.0 - self.rho)
    mom = mom * self.momentum + (g * self.lr) / torch.sqrt(ms + self.eps)
    return p - mom, ms, mom


class SlotView:
  """`optimizer.get_slot(var, name)`: a read view on one co-located state vector; plays the role of
  the `<param>/<opt>/<slot>` de.Variable of create_slots (PY/...optimizer.py:870-904)."""

  def __init__(self, var, field, init, name=None):
    self.var, self.field, self.init = var, field, init
    self.name = name if name is not None else "%s/slot%d" % (var.name, field)

  def size(self, index=None):
    """A slot vector lives in the row of its key: it has exactly the keys of the parameter."""
    return self.var.size(index)

  def lookup(self, keys):
    keys = torch.as_tensor(keys, device=self.var._primary)
    kp, perm, counts = self.var._partition(keys)
    outs = []
    for i, t in enumerate(self.var._tables):
      d = torch.full((self.var.dim,), self.init, dtype=self.var.value_dtype, device=t._device)
      outs.append(t._table.find(kp[i].to(t._device), d, field=self.field).to(self.var._primary))
    v = outs[0] if perm is None else device_ops.scatter_rows(torch.cat(outs, 0), perm)
    return v.reshape(tuple(keys.shape) + (self.var.dim,))

  def upsert(self, keys, values):
    """Write the state vector of `keys` (rows are created with the parameter's default if missing)."""
    keys = torch.as_tensor(keys, device=self.var._primary)
    kp, perm, counts = self.var._partition(keys)
    vp = self.var._split_rows(values.reshape(-1, self.var.dim), perm, counts)
    for i, t in enumerate(self.var._tables):
      if kp[i].numel():
        t._table.upsert(kp[i].to(t._device), vp[i].to(t._device), field=self.field)


class DynamicEmbeddingOptimizer:
  """`de.DynamicEmbeddingOptimizer(opt)` (PY/dynamic_embedding_optimizer.py:807-867)."""

  def __init__(self, opt, bp_v2=None, synchronous=False, exact_order=False):
    """exact_order=True sums duplicate gradients strictly in input order (bit-exact vs a sequential
    CPU unsorted_segment_sum) through unique + segment_sum + apply; the default uses the fused
    two-kernel path whose fixed summation tree is deterministic but not the sequential order."""
    if not isinstance(opt, _Opt):
      raise TypeError("optimizer must be one of tfra_amd.dynamic_embedding.optimizers.{SGD,Adam,Adagrad,Ftrl} or a "
                      "subclass of optimizers.Generic")
    self.opt = opt
    self.iterations = 0
    self.exact_order = exact_order

  @staticmethod
  def variable_kwargs(opt):
    """kwargs for de.Variable / get_variable so rows carry this optimizer's slots."""
    return {"aux_fields": len(opt.slots), "aux_init": opt.aux_init()}

  def get_slot(self, var, name):
    f = self.opt.slots.index(name) + 1
    return SlotView(var, f, self.opt.aux_init()[f - 1], "%s/%s/%s" % (var.name, type(self.opt).__name__, name))

  def get_slot_names(self):
    return list(self.opt.slots)

  def _check(self, var):
    if var.aux_fields < len(self.opt.slots):
      raise ValueError(
          "Variable %r was created with aux_fields=%d but %s needs %d co-located slot vectors: create it with "
          "**DynamicEmbeddingOptimizer.variable_kwargs(opt)" %
          (var.name, var.aux_fields, type(self.opt).__name__, len(self.opt.slots)))

  def begin_step(self):
    """Advance the global step ONCE and return the step's hyper-parameters (Adam's bias-corrected lr_t ...): pass them
    as `p` to every `apply_sparse` / `AllToAllEmbedding.apply_gradients` of the step when several tables share this
    optimizer — TF increments `iterations` once per apply_gradients, not once per variable."""
    self.iterations += 1
    return self.opt.params(self.iterations)

  def apply_gradients(self, grads_and_vars, name=None):
    """grads_and_vars: iterable of (grad[N,dim], TrainableWrapper) — one global step for all pairs."""
    p = self.begin_step()
    for grad, tw in grads_and_vars:
      if not isinstance(tw, TrainableWrapper):
        raise TypeError("expected the TrainableWrapper returned by embedding_lookup(..., return_trainable=True)")
      plan = tw.take_plan() if (self.opt.kind is not None and not self.exact_order) else None
      try:
        self.apply_sparse(tw.params, tw.ids, grad, p, plan=plan)
      finally:
        if plan is not None:
          from .variable import _release_plan
          _release_plan(tw.params, plan)

  def _apply_generic(self, var, ids, grad):
    """The reference's write-back sequence for an arbitrary rule (`Generic`)."""
    ids = torch.as_tensor(ids, device=var._primary).reshape(-1)
    grad = grad.reshape(-1, var.dim).to(torch.float32)
    n = ids.numel()
    if n == 0:
      return
    if getattr(var, "restrict_policy", None) is not None:
      var.restrict_policy.apply_update(ids)
    if var.dim % 4 == 0 and var.dim <= 256 and n <= (1 << 18) and not self.exact_order:
      uniq_buf, gsum, cnt = device_ops.reduce_by_key(ids, grad)
    else:
      uniq_buf, idx, cnt = device_ops.unique_no_sync(ids)
      gsum = device_ops.segment_sum(grad, idx, cnt, n)
    u = int(cnt.item())
    uniq, g = uniq_buf[:u], gsum[:u]
    views = [self.get_slot(var, name) for name in self.opt.slots]
    p = var.lookup(uniq).to(torch.float32)
    slots = [v.lookup(uniq).to(torch.float32) for v in views]
    new = self.opt.update(self.iterations, p, g, *slots)
    var.upsert(uniq, new[0].to(var.value_dtype))
    for v, s_new in zip(views, new[1:]):
      v.upsert(uniq, s_new.to(var.value_dtype))

  @staticmethod
  def can_plan(var, n):
    """The planned / two-kernel write-back covers one shard, fp32 rows with dim % 4 == 0, dim <= 256."""
    return (var.shard_num == 1 and not callable(var.initializer) and var.dim % 4 == 0 and var.dim <= 256 and
            n <= (1 << 18) and var.value_dtype == torch.float32)

  def plan(self, var, ids, plan=None):
    """Build (or rebuild) the id-only half of `apply_sparse(var, ids, ...)` on the current stream; pass the
    result as `apply_sparse(..., plan=plan)`.  Call it under `torch.cuda.stream(side_stream)` to overlap it
    with the lookup of the same ids or with the previous step."""
    from .table_ops import SparsePlan
    ids = torch.as_tensor(ids, device=var._primary).reshape(-1)
    if not self.can_plan(var, ids.numel()) or self.exact_order or self.opt.kind is None:
      raise ValueError("this variable / batch / optimizer takes the unique + segment_sum write-back, which has no plan")
    if plan is None:
      plan = SparsePlan(var._tables[0]._device, var.dim)
    return plan.build(ids)

  def apply_sparse(self, var, ids, grad, p=None, plan=None):
    """Sum gradients of duplicate ids, then one fused update per unique key and shard.  p: the step's parameters from
    `begin_step()`; None = this call IS the step (one variable per optimizer): the global step advances here."""
    if p is None:
      p = self.begin_step()
    self._check(var)
    if self.opt.kind is None:
      return self._apply_generic(var, ids, grad)
    if plan is not None:
      if getattr(var, "restrict_policy", None) is not None:
        var.restrict_policy.apply_update(plan.ids)
      t = var._tables[0]
      t._table.apply_planned(p, plan, grad.reshape(-1, var.dim), t._default_value.to(torch.float32))
      return
    ids = torch.as_tensor(ids, device=var._primary).reshape(-1)
    grad = grad.reshape(-1, var.dim).to(torch.float32)
    n = ids.numel()
    if n == 0:
      return
    if getattr(var, "restrict_policy", None) is not None:  # PY/embedding_weights.py:441-442
      var.restrict_policy.apply_update(ids)
    if (var.shard_num == 1 and not callable(var.initializer) and var.dim % 4 == 0 and var.dim <= 256 and not self.exact_order and
        var.value_dtype == torch.float32):
      # whole backward half in two kernels (tile reduce + bucket apply): no host sync, deterministic.  (More than 2^18
      # ids: the library reduces chunk by chunk and applies every key once — tfra_csr.hip: apply_sparse_big.)
      t = var._tables[0]
      t._table.apply_sparse(p, ids, grad, t._default_value.to(torch.float32))
      return
    if var.dim % 4 == 0 and var.dim <= 256 and n <= (1 << 18) and not self.exact_order:
      # sharded variables / callable initializers / half and bfloat16 rows: the same parallel, order-fixed duplicate reduction
      # (tile reduce + bucket merge, float32 sums), then one fused update per unique key and shard — on a half / bfloat16
      # table the rule runs in float32 on the up-cast row and slots and the results are rounded to the storage type once.  (unique +
      # segment_sum walks a segment sequentially: 9 ms for a Zipf batch whose hottest id repeats 24 000 times.)
      uniq_buf, gsum, cnt = device_ops.reduce_by_key(ids, grad)
    else:
      uniq_buf, idx, cnt = device_ops.unique_no_sync(ids)
      gsum = device_ops.segment_sum(grad, idx, cnt, n)
    if var.shard_num == 1 and not callable(var.initializer):
      # one shard: no partition, so the unique count never has to reach the host
      t = var._tables[0]
      t._table.apply_optimizer(p, uniq_buf, gsum, t._default_value.to(torch.float32), n_dev=cnt)
      return
    u = int(cnt.item())
    uniq, gsum = uniq_buf[:u], gsum[:u]
    kp, perm, counts = var._partition(uniq)
    gp = var._split_rows(gsum, perm, counts)
    for i, t in enumerate(var._tables):
      k = kp[i].to(t._device)
      if k.numel() == 0:
        continue
      dd = var._create_default_values_by_initializer(k.numel(), t._device)
      if dd is None:
        dd = t._default_value
      t._table.apply_optimizer(p, k, gp[i].to(t._device), dd.to(torch.float32))


class CapturedTrainStep:
  """lookup + sparse write-back of ONE single-shard Variable captured into a HIP graph.

  A training step on a 131 072-id batch is ~4 kernels of 10-30 us; launched eagerly the gaps between
  them cost as much as a kernel.  The step is stream-ordered end to end (no host sync, counts stay
  on the device), so it can be captured once and replayed: `step(ids, grads)` copies the batch into
  the static buffers, refreshes the device-side learning rate (Adam's lr_t changes every step) and
  replays.  The table must have room for the keys the replays will insert (`reserve`), because
  growth needs the host.
  """

  def __init__(self, var, optimizer, batch, reserve_slots=None):
    if var.shard_num != 1:
      raise ValueError("CapturedTrainStep needs a single-shard Variable")
    self.var, self.deo, self.batch = var, optimizer, int(batch)
    self.table = var.tables[0]._table
    dev = self.table.device
    self.ids = torch.zeros(self.batch, dtype=torch.int64, device=dev)
    self.grads = torch.zeros((self.batch, var.dim), dtype=torch.float32, device=dev)
    self.lr = torch.zeros(1, dtype=torch.float32, device=dev)
    self.out = None
    self.graph = None
    if reserve_slots:
      self.table.reserve(reserve_slots)

  def _params(self):
    self.deo.iterations += 1
    p = self.deo.opt.params(self.deo.iterations)
    self.lr.fill_(p.lr)
    p.d_lr = self.lr.data_ptr()
    return p

  def _body(self, p):
    self.out = self.var.lookup(self.ids)
    self.deo.apply_sparse(self.var, self.ids, self.grads, p)

  def capture(self, warmup_ids=None):
    """Two eager warm-up steps (sizes every scratch buffer) on a side stream, then the capture."""
    self.table.set_capture_safe(True)
    if warmup_ids is not None:
      self.ids.copy_(warmup_ids)
    side = torch.cuda.Stream(device=self.table.device)
    side.wait_stream(torch.cuda.current_stream(self.table.device))
    with torch.cuda.stream(side):
      for _ in range(2):
        self._body(self._params())
    torch.cuda.current_stream(self.table.device).wait_stream(side)
    torch.cuda.synchronize(self.table.device)
    p = self._params()
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph):
      self._body(p)
    return self

  def step(self, ids, grads=None):
    self.ids.copy_(ids.reshape(-1))
    if grads is not None:
      self.grads.copy_(grads.reshape(self.batch, -1))
    self._params()  # advances the global step and refreshes the device-side lr_t
    self.graph.replay()
    return self.out

  def close(self):
    self.table.set_capture_safe(False)


class PrefetchStep:
  """The two-stream training step driven by ONE C call per step (`tfra_table_step_prefetch`): main stream =
  lookup of batch i -> run sums following plan i -> fused update; second stream = plan of batch i+1.
  No graph, no Python between the launches, and normally no cross-queue event either (the C driver orders
  the streams through host-visible progress counters; NPLANS plans rotate so that a plan's buffers are
  idle long before they are rebuilt).

      ps = PrefetchStep(var, deo); ps.prime(first_ids)
      for ...: rows = ps.step(grads, next_ids)      # runs the staged batch, stages next_ids (None at the end)
  """
  NPLANS = 4

  def __init__(self, var, optimizer):
    from .table_ops import SparsePlan
    if var.shard_num != 1 or not DynamicEmbeddingOptimizer.can_plan(var, 1):
      raise ValueError("PrefetchStep needs a single-shard fp32 Variable with dim % 4 == 0, dim <= 256")
    optimizer._check(var)
    self.var, self.deo = var, optimizer
    self.t = var.tables[0]
    self.table = self.t._table
    self.dev = self.table.device
    self.plans = [SparsePlan(self.dev, var.dim) for _ in range(self.NPLANS)]
    self.ids = [None] * self.NPLANS
    self.default = self.t._default_value.to(device=self.dev, dtype=torch.float32).contiguous()
    self.side = torch.cuda.Stream(device=self.dev)
    self.cur = 0

  def prime(self, ids):
    """Stage the first batch: its plan is built on the current stream."""
    ids = torch.as_tensor(ids, device=self.dev).reshape(-1).to(torch.int64).contiguous()
    self.plans[self.cur].build(ids, sync=False)
    self.ids[self.cur] = ids
    return self

  def step(self, grads, next_ids=None, next_ids_ready=True):
    """next_ids_ready: the second stream does NOT wait for the current stream, so `next_ids` must be complete in
    memory when this is called (a batch staged by the input pipeline).  Pass False when it is still being produced
    on the current stream (device-side hashing, an async H2D copy): the second stream then waits for it (one
    cross-stream event per step)."""
    from .table_ops import _ptr
    cur = self.cur
    nxt_slot = (cur + 1) % self.NPLANS
    ids = self.ids[cur]
    n = ids.numel()
    self.deo.iterations += 1
    p = self.deo.opt.params(self.deo.iterations)
    grads = grads.reshape(n, self.var.dim)
    if grads.dtype != torch.float32 or not grads.is_contiguous():
      grads = grads.to(torch.float32).contiguous()
    if getattr(self.var, "restrict_policy", None) is not None and n:   # PY/embedding_weights.py:441-442
      self.var.restrict_policy.apply_update(ids)
    out = torch.empty((n, self.var.dim), dtype=torch.float32, device=self.dev)
    nxt = None
    if next_ids is not None:
      nxt = next_ids
      if not (torch.is_tensor(nxt) and nxt.dtype == torch.int64 and nxt.dim() == 1 and nxt.is_contiguous() and
              nxt.device == self.dev):
        nxt = torch.as_tensor(next_ids, device=self.dev).reshape(-1).to(torch.int64).contiguous()
      nxt.record_stream(self.side)
      self.ids[nxt_slot] = nxt                      # kept alive until its step has run
    main = torch.cuda.current_stream(self.dev)
    if nxt is not None and not next_ids_ready:
      self.side.wait_stream(main)
    _capi.call("tfra_table_step_prefetch", self.table._h, ctypes.byref(p), self.plans[cur]._h, _ptr(ids), _ptr(out),
               _ptr(self.default), _ptr(grads), _ptr(self.default), self.plans[nxt_slot]._h if nxt is not None else None,
               _ptr(nxt), 0 if nxt is None else nxt.numel(), ctypes.c_void_p(main.cuda_stream),
               ctypes.c_void_p(self.side.cuda_stream))
    self.cur = nxt_slot
    return out


class MultiTablePrefetchStep:
  """`PrefetchStep` for MANY tables on one GPU (a DLRM-style model: BASELINE configs[4] has 26): ONE C call per training
  step (`tfra_multi_step_prefetch`) hands every table's step to a small pool of host threads, tables on `streams` pairs
  of HIP streams round-robin, so that the kernels of different tables overlap instead of queueing behind one another on
  one stream.  Measured with 26 tables (226 M keys, 151 GB of rows, batch 131 072 per table): 1.35 ms per step with 4
  stream pairs issued by ONE thread, 1.67 ms with one PrefetchStep per table, 2.1 ms with plain calls; more host threads
  or more streams did not help (the step is bound by the GPU, not by launch overhead).  `optimizer` is shared: one global
  step per call (`begin_step`).

      ms = MultiTablePrefetchStep(vars, deo); ms.prime([ids_t for each table])
      for ...: rows = ms.step([grads_t ...], [next_ids_t ...])     # list of [n, dim_t] lookups of the staged batches
  """
  NPLANS = 4

  def __init__(self, variables, optimizer, streams=4, workers=1):
    from .table_ops import SparsePlan
    self.vars, self.deo, self.workers = list(variables), optimizer, int(workers)
    for v in self.vars:
      if v.shard_num != 1 or not DynamicEmbeddingOptimizer.can_plan(v, 1):
        raise ValueError("MultiTablePrefetchStep needs single-shard fp32 Variables with dim % 4 == 0, dim <= 256")
      optimizer._check(v)
    self.tables = [v.tables[0]._table for v in self.vars]
    self.dev = self.tables[0].device
    nt = len(self.vars)
    self.plans = [[SparsePlan(self.dev, v.dim) for _ in range(self.NPLANS)] for v in self.vars]
    self.ids = [[None] * self.NPLANS for _ in range(nt)]
    self.defaults = [v.tables[0]._default_value.to(device=self.dev, dtype=torch.float32).contiguous() for v in self.vars]
    ns = max(1, min(int(streams), nt))
    self.main = [torch.cuda.Stream(device=self.dev) for _ in range(ns)]
    self.side = [torch.cuda.Stream(device=self.dev) for _ in range(ns)]
    self.descs = (_capi.StepDesc * nt)()
    for i, d in enumerate(self.descs):
      d.struct_size = ctypes.sizeof(_capi.StepDesc)
      d.table = self.tables[i]._h
      d.find_default = self.defaults[i].data_ptr()
      d.param_default_row = self.defaults[i].data_ptr()
      d.scores = None
      d.main_stream = self.main[i % ns].cuda_stream
      d.side_stream = self.side[i % ns].cuda_stream
    self.cur = 0
    self._keep = None

  def prime(self, ids_list):
    for i, ids in enumerate(ids_list):
      ids = torch.as_tensor(ids, device=self.dev).reshape(-1).to(torch.int64).contiguous()
      self.plans[i][self.cur].build(ids, sync=False)
      self.ids[i][self.cur] = ids
    torch.cuda.current_stream(self.dev).synchronize()   # the plans were built on the current stream, the steps run on others
    return self

  def step(self, grads_list, next_ids_list=None):
    """grads_list[i]: [n_i, dim_i] float32 gradients of table i's staged batch; next_ids_list[i]: its next batch or None at the
    end.  Ordered against the caller's current stream both ways (inputs produced there are waited for, the returned rows are
    safe to consume there)."""
    cur, nxt_slot = self.cur, (self.cur + 1) % self.NPLANS
    p = self.deo.begin_step()
    outs, keep = [], [p]
    for i, v in enumerate(self.vars):
      ids = self.ids[i][cur]
      n = ids.numel()
      g = grads_list[i].reshape(n, v.dim)
      if g.dtype != torch.float32 or not g.is_contiguous():
        g = g.to(torch.float32).contiguous()
      out = torch.empty((n, v.dim), dtype=torch.float32, device=self.dev)
      nxt = None if next_ids_list is None else next_ids_list[i]
      if nxt is not None and not (torch.is_tensor(nxt) and nxt.dtype == torch.int64 and nxt.dim() == 1 and nxt.is_contiguous() and
                                  nxt.device == self.dev):
        nxt = torch.as_tensor(nxt, device=self.dev).reshape(-1).to(torch.int64).contiguous()
      if getattr(v, "restrict_policy", None) is not None and n:
        v.restrict_policy.apply_update(ids)
      d = self.descs[i]
      d.opt = ctypes.addressof(p)
      d.plan_cur = self.plans[i][cur]._h
      d.ids_cur = ids.data_ptr()
      d.rows_out = out.data_ptr()
      d.grads_or_values = g.data_ptr()
      d.plan_next = self.plans[i][nxt_slot]._h if nxt is not None else None
      d.ids_next = nxt.data_ptr() if nxt is not None else None
      d.n_next = 0 if nxt is None else nxt.numel()
      self.ids[i][nxt_slot] = nxt
      outs.append(out)
      keep.append(g)
    # The tables run on this driver's own streams.  What the caller produced on ITS stream — the gradients of backward, the
    # float32 / contiguous copies made above, the next ids — must be complete before any of them reads it, and what they
    # produce (the looked-up rows) before the caller's stream consumes it: one event each way, and the caching allocator is
    # told which streams use the buffers (it would otherwise hand `out` / `g` to someone else as soon as the caller drops them).
    caller = torch.cuda.current_stream(self.dev)
    ready = torch.cuda.Event()
    ready.record(caller)
    for st in self.main + self.side:
      st.wait_event(ready)
    for i in range(len(self.vars)):
      ms, ss = self.main[i % len(self.main)], self.side[i % len(self.side)]
      outs[i].record_stream(ms)
      keep[1 + i].record_stream(ms)
      if self.ids[i][nxt_slot] is not None:
        self.ids[i][nxt_slot].record_stream(ss)
        self.ids[i][nxt_slot].record_stream(ms)
    _capi.call("tfra_multi_step_prefetch", len(self.vars), ctypes.cast(self.descs, ctypes.c_void_p), self.workers)
    for st in self.main:
      done = torch.cuda.Event()
      done.record(st)
      caller.wait_event(done)   # `outs` are safe to use on the caller's stream
    self._keep = keep   # buffers of the step in flight stay alive until the next call
    self.cur = nxt_slot
    return outs

  def synchronize(self):
    for s in self.main + self.side:
      s.synchronize()


class PrefetchAssignStep:
  """`PrefetchStep` for a table WITHOUT a fused optimizer (any value dtype, bounded Hkv tables included): one C call
  per step (`tfra_table_step_prefetch_assign`) = lookup of batch i -> insert_or_assign of batch i's rows (ids may
  repeat: the last occurrence wins; on a table at max_capacity new keys evict by score) on the main stream, while the
  de-duplication plan of batch i+1 is built on a second stream.

      ps = PrefetchAssignStep(table); ps.prime(first_ids)
      for ...: rows = ps.step(values, next_ids)       # values [n, dim]: what batch i writes back
  """
  NPLANS = 4

  def __init__(self, table):
    from .table_ops import SparsePlan
    self.t = table
    self.table = table._table if hasattr(table, "_table") else table
    self.dev = self.table.device
    self.plans = [SparsePlan(self.dev, 0) for _ in range(self.NPLANS)]
    self.ids = [None] * self.NPLANS
    self.default = self.table._default_value
    self.side = torch.cuda.Stream(device=self.dev)   # (a lower priority for it changes nothing: measured)
    self.cur = 0
    self._main = None

  def prime(self, ids):
    ids = torch.as_tensor(ids, device=self.dev).reshape(-1).to(torch.int64).contiguous()
    self.plans[self.cur].build(ids, sync=False)
    self.ids[self.cur] = ids
    return self

  def step(self, values, next_ids=None, scores=None, lookup=True, next_ids_ready=True):
    """next_ids_ready: see PrefetchStep.step.  (The host side of a step is kept short — cached function pointer and stream
    handles, no per-call name lookups: with six kernel launches behind it this call IS the step time once the kernels are
    faster than the host can launch them.)"""
    cur = self.cur
    nxt_slot = (cur + 1) % self.NPLANS
    ids = self.ids[cur]
    n = ids.numel()
    dim, vdt, dev = self.table.dim, self.table.value_dtype, self.dev
    if values.dtype != vdt or not values.is_contiguous() or values.shape != (n, dim):
      values = values.reshape(n, dim).to(vdt).contiguous()   # (raises on a size mismatch: the kernels index row `last` of [n, dim])
    out = torch.empty((n, dim), dtype=vdt, device=dev) if lookup else None
    nxt = None
    if self._main is None:
      main = torch.cuda.current_stream(dev)
      self._main, self._main_h, self._side_h = main, ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(self.side.cuda_stream)
      self._fn = _capi.lib().tfra_table_step_prefetch_assign
      self._default_p = ctypes.c_void_p(self.default.data_ptr())
    elif torch.cuda.current_stream(dev) != self._main:
      main = torch.cuda.current_stream(dev)
      self._main, self._main_h = main, ctypes.c_void_p(main.cuda_stream)
    if next_ids is not None:
      nxt = next_ids
      if not (torch.is_tensor(nxt) and nxt.dtype == torch.int64 and nxt.dim() == 1 and nxt.is_contiguous() and nxt.device == dev):
        nxt = torch.as_tensor(next_ids, device=dev).reshape(-1).to(torch.int64).contiguous()
      nxt.record_stream(self.side)
      self.ids[nxt_slot] = nxt
      if not next_ids_ready:
        self.side.wait_stream(self._main)
    sp = None
    if scores is not None:
      scores = scores.to(dev, torch.int64).contiguous()
      sp = ctypes.c_void_p(scores.data_ptr())
    _capi.check(self._fn(self.table._h, self.plans[cur]._h, ctypes.c_void_p(ids.data_ptr()),
                         ctypes.c_void_p(out.data_ptr()) if lookup else None, self._default_p, ctypes.c_void_p(values.data_ptr()), sp,
                         self.plans[nxt_slot]._h if nxt is not None else None, ctypes.c_void_p(nxt.data_ptr()) if nxt is not None else None,
                         0 if nxt is None else nxt.numel(), self._main_h, self._side_h))
    self.cur = nxt_slot
    return out


def assign_step_driver_for(new_key_ratio):
  """Which of the table's two step drivers a lookup + insert_or_assign stream should use, by RULE: 'overlapped_step'
  (OverlapAssignStep: lookup i+1 and write-back i in one launch, shared ids forwarded) when at most a tenth of a batch's ids are
  never-seen keys; 'look_ahead' (PrefetchAssignStep) beyond — every never-seen key on a full bounded table is an eviction, the
  write-back becomes the long pole and runs faster as kernels of its own with the whole chip and its own register budget.
  (Where the two cross on the metric's table, profiles/r06_m1b_new_key_ratio_sweep.log: 0 % never-seen ids 21.5 vs 36.3 us per step,
  1 % 29.2 vs 37.3, 5 % 43.6 vs 43.1, 10 % 46.8 vs 46.9, 25 % 58.5 vs 53.2 — until round 6 the rule said a quarter.)"""
  return "overlapped_step" if float(new_key_ratio) <= 0.10 else "look_ahead"


class _LookAheadAssignStep:
  """`PrefetchAssignStep` behind `OverlapAssignStep`'s call signature (prime / step(values, next_ids, next2_ids, return_exists) / flush),
  so that code written against one driver runs on the other (assign_step_for)."""

  def __init__(self, table):
    self._ps = PrefetchAssignStep(table)
    self._ids = None

  def prime(self, ids):
    self._ps.prime(ids)
    self._ids = self._ps.ids[self._ps.cur]
    return self

  def step(self, values, next_ids=None, next2_ids=None, return_exists=False):
    if self._ids is None:
      raise RuntimeError("assign step: no batch is primed — call prime(ids) first")
    ids = self._ids
    if return_exists:   # the look-ahead call returns rows only: the flags come from a find of their own, in front of the write-back
      rows, ex = self._ps.table.find(ids, return_exists=True)
      self._ps.step(values, next_ids, lookup=False)
    else:
      rows, ex = self._ps.step(values, next_ids), None
    self._ids = self._ps.ids[self._ps.cur] if next_ids is not None else None
    return (rows, ex) if return_exists else rows

  def flush(self):   # nothing is deferred: a step has written its batch back when it returns
    pass


def assign_step_for(table, new_key_ratio=0.0):
  """The step driver object assign_step_driver_for() names, behind one interface: prime(ids) / step(values, next_ids, next2_ids,
  return_exists) / flush()."""
  return OverlapAssignStep(table) if assign_step_driver_for(new_key_ratio) == "overlapped_step" else _LookAheadAssignStep(table)


class OverlapAssignStep:
  """The overlapped step (`tfra_table_step_overlap`, csrc/tfra_step_impl.h): same use as `PrefetchAssignStep` —

      os_ = OverlapAssignStep(table).prime(first_ids)
      for ...: rows = os_.step(values, next_ids, next2_ids)   # rows = lookup(batch i); values [n, dim] = what batch i writes back
      os_.flush()                                             # before the table is used any other way

  — and the same results as lookup(i); insert_or_assign(i); lookup(i+1); ... one after the other, but the write-back of
  batch i runs in the SAME kernel launch as the lookup of batch i+1: ids the two batches share are served from `values` (which
  must therefore stay unchanged until the next step has run), a key the write-back evicts although the next lookup asks for it
  is corrected afterwards.  ONE launch per step on one stream, no second stream, no host synchronisation.  `make_run()` enqueues
  many steps with ONE host call."""

  def __init__(self, table):
    self.t = table
    self.table = table._table if hasattr(table, "_table") else table
    self.dev = self.table.device
    self._h = ctypes.c_void_p()
    _capi.call("tfra_step_driver_create", self.table._h, ctypes.byref(self._h))
    self.default = self.table._default_value
    if self.default.dtype != self.table.value_dtype or not self.default.is_contiguous():
      self.default = self.default.to(self.table.value_dtype).contiguous()
    self._ids = None
    self._pending = None      # (ids, values) of the batch still to be written back: kept alive until it has been
    self._keep = None
    self._keep2 = None
    self._fn = _capi.lib().tfra_table_step_overlap
    self._default_p = ctypes.c_void_p(self.default.data_ptr())

  def __del__(self):
    try:
      if self._h:
        _capi.lib().tfra_step_driver_destroy(self._h)
        self._h = None
    except Exception:
      pass

  def _as_ids(self, ids):
    if torch.is_tensor(ids) and ids.dtype == torch.int64 and ids.dim() == 1 and ids.is_contiguous() and ids.device == self.dev:
      return ids
    return torch.as_tensor(ids, device=self.dev).reshape(-1).to(torch.int64).contiguous()

  def prime(self, ids):
    self._ids = self._as_ids(ids)
    return self

  def step(self, values, next_ids=None, next2_ids=None, return_exists=False):
    """next_ids / next2_ids: the ids of the next step and of the one after it (an input pipeline knows them): with both, the
    de-duplication plan of a batch is built inside the step launches, without atomics; with next_ids only, by a launch of its
    own per step; with neither, in front of the next step."""
    from .table_ops import _stream
    ids = self._ids
    if ids is None:
      raise RuntimeError("OverlapAssignStep.step: no batch is primed — call prime(ids) first (and again after a step that "
                         "announced no next_ids, or after a run made without ids_after)")
    n = ids.numel()
    dim, vdt, dev = self.table.dim, self.table.value_dtype, self.dev
    if values.dtype != vdt or not values.is_contiguous() or values.shape != (n, dim):
      values = values.reshape(n, dim).to(vdt).contiguous()
    out = torch.empty((n, dim), dtype=vdt, device=dev)
    ex = torch.empty(n, dtype=torch.bool, device=dev) if return_exists else None
    nxt = self._as_ids(next_ids) if next_ids is not None else None
    nx2 = self._as_ids(next2_ids) if next2_ids is not None else None
    prev = self._pending
    _capi.check(self._fn(self._h, n, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                         ctypes.c_void_p(ex.data_ptr()) if ex is not None else None, self._default_p, 0,
                         ctypes.c_void_p(prev[1].data_ptr()) if prev is not None else None, None,
                         0 if nxt is None else nxt.numel(), ctypes.c_void_p(nxt.data_ptr()) if nxt is not None else None,
                         0 if nx2 is None else nx2.numel(), ctypes.c_void_p(nx2.data_ptr()) if nx2 is not None else None, _stream(dev)))
    self._keep = (prev, self._keep2)   # their buffers are read by the launch just enqueued
    self._keep2 = nx2
    self._pending = (ids, values)
    self._ids = nxt
    return (out, ex) if return_exists else out

  def flush(self):
    from .table_ops import _stream
    if self._pending is not None:
      _capi.call("tfra_table_step_overlap_flush", self._h, ctypes.c_void_p(self._pending[1].data_ptr()), None, _stream(self.dev))
      self._keep = self._pending
      self._pending = None

  def stats(self):
    a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
    dc = (ctypes.c_uint32 * 3)()
    why = ctypes.c_uint32()
    pb = (ctypes.c_uint64 * 2)()
    _capi.call("tfra_step_driver_stats", self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), dc, ctypes.byref(why), pb)
    ll = ctypes.c_uint64()
    _capi.call("tfra_step_driver_lookups_listed", self._h, ctypes.byref(ll))
    return {"overlapped": a.value, "sequential": b.value, "pending": bool(c.value), "deferred_evictions": dc[0],
            "victims_noted": dc[1], "rows_corrected": dc[2], "why_sequential": why.value, "plans_built_in_launch": pb[0],
            "plans_built_in_front": pb[1], "lookups_listed": ll.value}

  def time_kernels(self, steps):
    """HIP events around the launch of each of the next `steps` overlapped steps (on their stream); read with kernel_times()."""
    _capi.call("tfra_step_driver_time_kernels", self._h, int(steps))

  def kernel_times(self):
    a, b, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_size_t()
    _capi.call("tfra_step_driver_kernel_times", self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n))
    return {"step_kernel_us": a.value, "rest_kernel_us": b.value, "steps": n.value}

  def timing(self):
    """tuning (TFRA_STEP_VARIANT & 16): per launch and role — build, scatter, write-back, lookup, tail, map — (first block start, last
    block end, median block duration, 95th percentile) in us, starts / ends since the launch's first block (None: the role did not run)"""
    buf = (ctypes.c_uint64 * (64 * 6 * 4))()
    _capi.call("tfra_step_driver_timing", self._h, buf)
    none = 2 ** 64 - 1
    out = []
    for k in range(64):
      w = [tuple(buf[(k * 6 + r) * 4 + j] for j in range(4)) for r in range(6)]
      starts = [x[0] for x in w if x[0] != none]
      if not starts:
        continue
      t0 = min(starts)
      out.append([None if x[0] == none else ((x[0] - t0) / 100.0, (x[1] - t0) / 100.0, x[2] / 100.0, x[3] / 100.0) for x in w])
    return out

  def make_run(self, ids_list, values_list, outs, ids_after=None, values_before=None, ids_after2=None):
    """Pre-builds the argument array of `tfra_table_steps_overlap` for the steps (ids_list[k], values_list[k]) -> outs[k]:
    returns a callable that enqueues all of them with ONE host call.  values_before = the values of the batch pending when
    the run starts (None: nothing pending); ids_after / ids_after2 = the ids of the two steps behind the run (their plans are
    built / started by the run's last launches).  The caller keeps every tensor alive and unchanged until the run has executed."""
    from .table_ops import _stream
    m = len(ids_list)
    arr = (_capi.OverlapStep * m)()
    for k in range(m):
      q = arr[k]
      q.struct_size = ctypes.sizeof(_capi.OverlapStep)
      q.default_is_full = 0
      q.n = ids_list[k].numel()
      q.ids = ids_list[k].data_ptr()
      q.rows_out = outs[k].data_ptr()
      q.exists_out = None
      q.defaults = self.default.data_ptr()
      vp = values_before if k == 0 else values_list[k - 1]
      q.values_prev = vp.data_ptr() if vp is not None else None
      q.scores_prev = None
      nx = ids_list[k + 1] if k + 1 < m else ids_after
      q.n_next = nx.numel() if nx is not None else 0
      q.ids_next = nx.data_ptr() if nx is not None else None
      n2 = ids_list[k + 2] if k + 2 < m else (ids_after if k + 2 == m else ids_after2)
      q.n_next2 = n2.numel() if n2 is not None else 0
      q.ids_next2 = n2.data_ptr() if n2 is not None else None
    fn = _capi.lib().tfra_table_steps_overlap
    h, dev = self._h, self.dev
    last = (ids_list[-1], values_list[-1])

    def run():
      _capi.check(fn(h, m, arr, _stream(dev)))
      self._pending = last
      self._ids = ids_after
    run._keep = (arr, ids_list, values_list, outs, ids_after, values_before, ids_after2)
    return run
