"""`Variable`, `get_variable`, `embedding_lookup*` — host-side mirror of the reference API.

PY = /root/reference/tensorflow_recommenders_addons/dynamic_embedding/python/ops
  Variable ............... PY/dynamic_embedding_variable.py:453-1262
  default_partition_fn ... PY/dynamic_embedding_variable.py:165-197
  get_variable ........... PY/dynamic_embedding_variable.py:1265-1359
  embedding_lookup ....... PY/dynamic_embedding_variable.py:1362-1530
  embedding_lookup_unique / _sparse / safe_..._sparse  PY/dynamic_embedding_ops.py:64-430
  TrainableWrapper ....... PY/embedding_weights.py:38-540 (prefetch_values / update_op)

A logical table = N physical tables (`devices=[...]`, one shard per entry); every call is
partition -> per-shard table op -> stitch, the partition/stitch being device kernels
(tfra_partition / tfra_scatter_rows) instead of tf.dynamic_partition/dynamic_stitch.
"""
import torch

from . import device_ops
from .table_ops import CuckooHashTable, HkvHashTable, HkvEvictStrategy, _as_device


def default_partition_fn(keys, shard_num):
  """PY/dynamic_embedding_variable.py:165-197 (int64 keys, accelerator build):
  ``int32(key & 0x7fffffff) % shard_num``.  Kept as a python callable for API parity; the
  Variable recognises it and runs the fused device partition instead."""
  if shard_num <= 1:
    return torch.zeros(keys.shape, dtype=torch.int32, device=keys.device)
  return ((keys & 0x7FFFFFFF).to(torch.int32) % shard_num).to(torch.int32)


class KVCreator:
  """PY/dynamic_embedding_creator.py:36-78"""

  def __init__(self, config=None):
    self.config = config

  def create(self, **kw):
    raise NotImplementedError


class CuckooHashTableConfig:
  """PY/dynamic_embedding_creator.py:80-88 (empty in the reference too)."""


class CuckooHashTableCreator(KVCreator):
  """PY/dynamic_embedding_creator.py:91-138"""

  def create(self, key_dtype=None, value_dtype=None, default_value=None, name=None, checkpoint=None, init_size=None,
             config=None, device=None, shard_saveable_object_fn=None, **kw):
    return CuckooHashTable(key_dtype=key_dtype, value_dtype=value_dtype, default_value=default_value, name=name,
                           checkpoint=checkpoint, init_size=init_size or 0, config=config or self.config, device=device,
                           **kw)


class HkvHashTableConfig:
  """PY/dynamic_embedding_creator.py:149-169"""

  def __init__(self, init_capacity=1024 * 1024, max_capacity=1024 * 1024, max_hbm_for_values=1024 * 1024 * 1024,
               evict_strategy=HkvEvictStrategy.LRU, step_per_epoch=0, gen_scores_fn=None, reserved_key_start_bit=0):
    self.init_capacity = init_capacity
    self.max_capacity = max_capacity
    self.max_hbm_for_values = max_hbm_for_values
    self.evict_strategy = evict_strategy
    self.step_per_epoch = step_per_epoch
    self.gen_scores_fn = gen_scores_fn
    self.reserved_key_start_bit = reserved_key_start_bit


class HkvHashTableCreator(KVCreator):
  """PY/dynamic_embedding_creator.py:172-230"""

  def create(self, key_dtype=None, value_dtype=None, default_value=None, name=None, checkpoint=None, init_size=None,
             config=None, device=None, shard_saveable_object_fn=None, **kw):
    cfg = config or self.config or HkvHashTableConfig()
    return HkvHashTable(key_dtype=key_dtype, value_dtype=value_dtype, default_value=default_value, name=name,
                        checkpoint=checkpoint, config=cfg, device=device, **kw)


class Variable:
  """A sharded dynamic-embedding table (PY/dynamic_embedding_variable.py:453-692)."""

  def __init__(self, key_dtype=torch.int64, value_dtype=torch.float32, dim=1, devices=None,
               partitioner=default_partition_fn, shared_name=None, name="DynamicEmbedding_Variable", initializer=None,
               trainable=True, checkpoint=True, init_size=0, kv_creator=None, restrict_policy=None, bp_v2=False,
               short_file_name=False, aux_fields=0, aux_init=(0.0, 0.0, 0.0, 0.0)):
    self.key_dtype = key_dtype
    self.value_dtype = value_dtype
    self.dim = int(dim)
    self.bp_v2 = bp_v2
    self.name = name
    self.trainable = trainable
    self.checkpoint = checkpoint
    self.aux_fields = aux_fields
    self.devices = list(devices) if devices else ["cuda:%d" % torch.cuda.current_device()]
    self.shard_num = len(self.devices)
    self.partition_fn = partitioner
    self.init_size = int(init_size / self.shard_num)  # PY/..._variable.py:597
    self.initializer = initializer
    self.kv_creator = kv_creator if kv_creator else CuckooHashTableCreator()
    # static default row: first element of the initializer (PY/..._variable.py:723-765)
    if initializer is None:
      static_default = 0
    elif callable(initializer):
      static_default = initializer([1, self.dim]).reshape(-1)[0].item()
    else:
      static_default = torch.as_tensor(initializer).reshape(-1)[0].item()
    self._static_default = static_default
    default_value = torch.full((self.dim,), static_default, dtype=value_dtype)
    self._tables = []
    for idx, dev in enumerate(self.devices):
      self._tables.append(
          self.kv_creator.create(key_dtype=key_dtype, value_dtype=value_dtype, default_value=default_value,
                                 name=self._make_name(idx), checkpoint=checkpoint, init_size=self.init_size, device=dev,
                                 dim=self.dim, aux_fields=aux_fields, aux_init=aux_init))
    self._primary = _as_device(self.devices[0])
    # PY/dynamic_embedding_variable.py:604-611: the policy CLASS is passed, instantiated on this variable
    if restrict_policy is not None:
      from .restrict_policies import RestrictPolicy
      if not (isinstance(restrict_policy, type) and issubclass(restrict_policy, RestrictPolicy)):
        raise TypeError("restrict_policy must be subclass of RestrictPolicy.")
      self._restrict_policy = restrict_policy(self)
    else:
      self._restrict_policy = None

  @property
  def restrict_policy(self):
    return self._restrict_policy

  def get_slot_variables(self, optimizer):
    """PY/dynamic_embedding_variable.py:1157-1200: the optimizer's state variables of this parameter, sorted by
    name (here: views on the co-located state vectors of the rows)."""
    return sorted((optimizer.get_slot(self, n) for n in optimizer.get_slot_names()), key=lambda v: v.name)

  def restrict(self, num_reserved, **kwargs):
    """PY/dynamic_embedding_variable.py:857-874: no-op without a policy."""
    if self._restrict_policy is not None:
      return self._restrict_policy.apply_restriction(num_reserved, **kwargs)
    return None

  def _make_name(self, table_idx):
    """PY/dynamic_embedding_variable.py:768-770"""
    return "{}_mht_{}of{}".format(self.name.replace("/", "_"), table_idx + 1, self.shard_num)

  @property
  def tables(self):
    return self._tables

  # ---- partition / stitch ------------------------------------------------------------------
  def _partition(self, keys):
    """-> (keys_per_shard, perm, counts_host). perm is None for 1 shard."""
    flat = keys.reshape(-1)
    if self.shard_num <= 1:
      return [flat], None, [flat.numel()]
    if self.partition_fn is default_partition_fn:
      if flat.dtype == torch.int32:   # the partition kernel reads int64 keys; sign extension keeps (key & 0x7fffffff) % n
        owner_major, perm, counts = device_ops.partition(flat.to(torch.int64), self.shard_num, device_ops.PARTITION_MASK_MOD)
        owner_major = owner_major.to(torch.int32)
      else:
        owner_major, perm, counts = device_ops.partition(flat, self.shard_num, device_ops.PARTITION_MASK_MOD)
    else:
      owner = self.partition_fn(flat, self.shard_num)
      perm, counts = device_ops.partition_by_owner(owner, self.shard_num)
      owner_major = device_ops.gather_rows(flat.reshape(-1, 1), perm).reshape(-1)
    c = counts.tolist()  # data-dependent shapes: one host read, like tf.dynamic_partition
    return list(torch.split(owner_major, c)), perm, c

  def _split_rows(self, rows, perm, counts):
    if perm is None:
      return [rows]
    return list(torch.split(device_ops.gather_rows(rows, perm), counts))

  # ---- table ops ---------------------------------------------------------------------------
  def upsert(self, keys, values, name=None):
    """PY/dynamic_embedding_variable.py:772-804"""
    keys = torch.as_tensor(keys, device=self._primary)
    values = torch.as_tensor(values, device=self._primary)
    want = tuple(keys.shape) + (self.dim,)
    if tuple(values.shape) != want:
      raise ValueError("Expected shape %s for values, got %s" % (list(want), list(values.shape)))
    kp, perm, counts = self._partition(keys)
    vp = self._split_rows(values.reshape(-1, self.dim), perm, counts)
    for i, t in enumerate(self._tables):
      t.insert(kp[i].to(t._device), vp[i].to(t._device))

  def accum(self, keys, old_values, new_values, exists, name=None):
    """PY/dynamic_embedding_variable.py:806-855: where(exists, new-old, new) then per-shard accum."""
    keys = torch.as_tensor(keys, device=self._primary)
    exists = torch.as_tensor(exists, device=self._primary).reshape(-1).to(torch.bool)
    old_values = old_values.reshape(-1, self.dim)
    new_values = new_values.reshape(-1, self.dim)
    vod = torch.where(exists[:, None], new_values - old_values, new_values)
    kp, perm, counts = self._partition(keys)
    vp = self._split_rows(vod, perm, counts)
    ep = [exists] if perm is None else list(
        torch.split(device_ops.gather_rows(exists.reshape(-1, 1).to(torch.uint8), perm).reshape(-1).to(torch.bool),
                    counts))
    for i, t in enumerate(self._tables):
      t.accum(kp[i].to(t._device), vp[i].to(t._device), ep[i].to(t._device))

  def remove(self, keys, name=None):
    """PY/dynamic_embedding_variable.py:877-902"""
    kp, _, _ = self._partition(torch.as_tensor(keys, device=self._primary))
    for i, t in enumerate(self._tables):
      t.remove(kp[i].to(t._device))

  def clear(self, name=None):
    for t in self._tables:
      t.clear()

  def _create_default_values_by_initializer(self, n, device):
    """PY/dynamic_embedding_variable.py:919-931: full-size [n,dim] defaults from the initializer."""
    if self.initializer is None or not callable(self.initializer):
      return None
    return self.initializer([n, self.dim]).to(device=device, dtype=self.value_dtype)

  def lookup(self, keys, return_exists=False, name=None):
    """PY/dynamic_embedding_variable.py:933-986"""
    keys = torch.as_tensor(keys, device=self._primary)
    kp, perm, counts = self._partition(keys)
    vals, exs = [], []
    for i, t in enumerate(self._tables):
      k = kp[i].to(t._device)
      dd = self._create_default_values_by_initializer(k.numel(), t._device)
      r = t.lookup(k, dynamic_default_values=dd, return_exists=return_exists)
      if return_exists:
        vals.append(r[0].to(self._primary))
        exs.append(r[1].to(self._primary))
      else:
        vals.append(r.to(self._primary))
    if perm is None:
      v = vals[0]
      e = exs[0] if return_exists else None
    else:
      v = device_ops.scatter_rows(torch.cat(vals, 0), perm)
      e = None
      if return_exists:
        e = device_ops.scatter_rows(torch.cat(exs, 0).reshape(-1, 1).to(torch.uint8), perm).reshape(-1).to(torch.bool)
    v = v.reshape(tuple(keys.shape) + (self.dim,))
    if return_exists:
      return v, e.reshape(keys.shape)
    return v

  def export(self, name=None):
    """PY/dynamic_embedding_variable.py:988-1007"""
    ks, vs = [], []
    for t in self._tables:
      k, v = t.export()
      ks.append(k.to(self._primary))
      vs.append(v.to(self._primary))
    return torch.cat(ks, 0), torch.cat(vs, 0)

  def size(self, index=None, name=None):
    """PY/dynamic_embedding_variable.py:1133-1155"""
    if index is not None:
      return self._tables[index].size()
    return torch.stack([t.size().to(self._primary) for t in self._tables]).sum()

  def _slot_file_names(self, optimizer):
    """{field: base file name} of the co-located state vectors.  With the optimizer: the reference's slot-variable
    names `<param>/<opt>/<slot>` (create_slots, PY/dynamic_embedding_optimizer.py:870-904; each is a table of its own
    there and is checkpointed as `<param>_<opt>_<slot>_mht_<i>of<N>`), else `<param>_slot<f>`."""
    if optimizer is not None:
      return {v.field: v.name.replace("/", "_") for v in self.get_slot_variables(optimizer)}
    return {f: "%s_slot%d" % (self.name.replace("/", "_"), f) for f in range(1, self.aux_fields + 1)}

  def save_to_file_system(self, dirpath, proc_size=1, proc_rank=0, buffer_size=4194304, optimizer=None):
    """Per-shard files `<name>_mht_<i>of<N>[_rank<r>_size<s>]-keys/-values`
    (PY/dynamic_embedding_variable.py:1009-1060) — and the same for every optimizer slot of the rows (the reference
    saves its slot variables as tables of their own; not saving them would silently reset Adam's m / v, Adagrad's and
    FTRL's accumulators on restore).  The optimizer's step count (`DynamicEmbeddingOptimizer.iterations`, Adam's bias
    correction) is the OPTIMIZER's state, not the variable's: the reference keeps it in the Keras optimizer's own
    checkpoint, and so must the caller here (`deo.iterations` is a plain int)."""
    import os
    suffix = "_rank{}_size{}".format(proc_rank, proc_size) if proc_size > 1 else ""
    slots = self._slot_file_names(optimizer)
    for idx, t in enumerate(self._tables):
      t.save_to_file_system(dirpath, file_name=self._make_name(idx) + suffix, dirpath_env=None, buffer_size=buffer_size)
      for f, base in slots.items():
        t._table.save(os.path.join(dirpath, "{}_mht_{}of{}{}".format(base, idx + 1, self.shard_num, suffix)), buffer_size,
                      field=f)

  def load_from_file_system(self, dirpath, proc_size=1, proc_rank=0, buffer_size=4194304, optimizer=None):
    """Reload; when the shard count changed, every `_mht_` file is re-read and re-partitioned
    through partition_fn (PY/dynamic_embedding_variable.py:200-450, 1062-1131).  Slot files written by
    save_to_file_system are restored into the rows' state vectors after the embedding itself."""
    import os
    import numpy as np
    own = self.name.replace("/", "_") + "_mht_"
    listing = os.listdir(dirpath)
    files = sorted(f[:-len("-keys")] for f in listing if f.endswith("-keys") and f.startswith(own))
    slots = self._slot_file_names(optimizer)

    def slot_files(base):
      return sorted(f[:-len("-keys")] for f in listing if f.endswith("-keys") and f.startswith(base + "_mht_"))

    # The slot files are named after the optimizer's slot variables when an optimizer is passed, `<param>_slot<f>` when not.
    # Saving one way and restoring the other must not silently reset Adam's m / v or FTRL's accumulators: when the expected
    # names are absent, the other scheme is looked for — by EXACT base name only (`<param>_<OptClass>_<slot>` for the known
    # optimizer classes, mapped to the field by slot NAME; or `<param>_slot<f>`), never by prefix: a sibling variable
    # `<param>_2` or `<param>_user` in the same directory is not this variable's state.  State files of this variable that
    # match neither scheme completely are an error, not a skip.
    if self.aux_fields:
      prefix = self.name.replace("/", "_") + "_"
      siblings = {n.replace("/", "_") for n in _VARIABLES if n != self.name}
      missing = [f for f, base in slots.items() if not slot_files(base)]
      if missing:
        generic = {f: prefix + "slot%d" % f for f in slots}
        layouts = {cls: tuple(sl) for cls, sl in _known_slot_layouts().items() if len(sl) == self.aux_fields}
        named = {cls: {i + 1: prefix + cls + "_" + sname for i, sname in enumerate(sl)} for cls, sl in layouts.items()}
        named = {cls: m for cls, m in named.items() if not any(b in siblings for b in m.values())}
        if optimizer is None:
          complete = [cls for cls, m in named.items() if all(slot_files(b) for b in m.values())]
          if len(complete) == 1 and len(missing) == len(slots):
            slots = dict(named[complete[0]])
        else:
          if all(slot_files(generic[f]) for f in missing) and not any(generic[f] in siblings for f in missing):
            for f in missing:
              slots[f] = generic[f]
        still = [f for f, base in slots.items() if not slot_files(base)]
        if still:
          # is there ANY state file that can only be this variable's?  (exact bases of either scheme)
          cands = set(generic.values())
          for m in named.values():
            cands.update(m.values())
          found = sorted(b for b in cands if b not in siblings and slot_files(b))
          if found:
            raise ValueError("load_from_file_system: %r holds optimizer-state files %s of variable %r, but they do not cover slot "
                             "field(s) %s (saved with another optimizer?); pass the optimizer the checkpoint was saved with"
                             % (dirpath, found, self.name, still))

    same = all(self._make_name(i) in files for i in range(self.shard_num)) and len(files) == self.shard_num
    if same and proc_size == 1:
      for idx, t in enumerate(self._tables):
        t.load_from_file_system(dirpath, file_name=self._make_name(idx), dirpath_env=None, buffer_size=buffer_size)
        for f, base in slots.items():
          p = os.path.join(dirpath, "{}_mht_{}of{}".format(base, idx + 1, self.shard_num))
          if os.path.exists(p + "-keys"):
            t._table.load(p, buffer_size, field=f)
      return
    self.clear()
    for fn in files:
      keys = np.fromfile(os.path.join(dirpath, fn + "-keys"), dtype=np.int64)
      vals = torch.from_numpy(np.fromfile(os.path.join(dirpath, fn + "-values"), dtype=np.uint8)).view(
          self.value_dtype).reshape(-1, self.dim)
      self.upsert(torch.from_numpy(keys).to(self._primary), vals.to(self._primary))
    for f, base in slots.items():   # re-sharded restore of the state vectors: through the same partitioner
      for fn in slot_files(base):
        keys = torch.from_numpy(np.fromfile(os.path.join(dirpath, fn + "-keys"), dtype=np.int64)).to(self._primary)
        vals = torch.from_numpy(np.fromfile(os.path.join(dirpath, fn + "-values"), dtype=np.uint8)).view(
            self.value_dtype).reshape(-1, self.dim).to(self._primary)
        kp, perm, counts = self._partition(keys)
        vp = self._split_rows(vals, perm, counts)
        for i, t in enumerate(self._tables):
          if kp[i].numel():
            t._table.upsert(kp[i].to(t._device), vp[i].to(t._device), field=f)


_VARIABLES = {}


def _known_slot_layouts():
  """{optimizer class name: slot names in field order} of the optimizers this package ships (the `<opt>` and `<slot>` parts of
  the reference's slot-variable names `<param>/<opt>/<slot>`, PY/dynamic_embedding_optimizer.py:870-904)."""
  from . import optimizer as _o
  out = {}
  for cls in (_o.Adam, _o.Adagrad, _o.Ftrl, _o.Momentum, _o.RMSProp):
    sl = cls.slots if cls.slots else cls().slots
    if sl:
      out[cls.__name__] = tuple(sl)
  return out


def get_variable(name, key_dtype=torch.int64, value_dtype=torch.float32, dim=1, devices=None,
                 partitioner=default_partition_fn, shared_name="get_variable", initializer=None, trainable=True,
                 checkpoint=True, init_size=0, kv_creator=None, restrict_policy=None, bp_v2=False, **kw):
  """PY/dynamic_embedding_variable.py:1265-1359: create-or-reuse by name."""
  if name in _VARIABLES:
    return _VARIABLES[name]
  v = Variable(key_dtype=key_dtype, value_dtype=value_dtype, dim=dim, devices=devices, partitioner=partitioner,
               shared_name=shared_name, name=name, initializer=initializer, trainable=trainable, checkpoint=checkpoint,
               init_size=init_size, kv_creator=kv_creator, restrict_policy=restrict_policy, bp_v2=bp_v2, **kw)
  _VARIABLES[name] = v
  return v


PLAN_AT_LOOKUP_MIN_IDS = 8192   # below this the write-back plan is not worth a second stream
PLAN_POOL_MAX = 8


def _plan_at_lookup(params, ids):
  """The id-only half of the optimizer write-back of `ids` (the de-duplication plan, SparsePlan), started on the
  Variable's second stream at LOOKUP time: the ids of a training step are known when it looks them up (the reference
  keeps them in the TrainableWrapper, PY/embedding_weights.py:38-120) and the plan needs nothing else, so it builds while
  the lookup, the model's forward and backward run — no look-ahead into the next batch is needed.  Returns None when
  the write-back of this variable / batch is not the planned one."""
  from .optimizer import DynamicEmbeddingOptimizer
  from .table_ops import SparsePlan
  n = ids.numel()
  if n < PLAN_AT_LOOKUP_MIN_IDS or params.bp_v2 or not DynamicEmbeddingOptimizer.can_plan(params, n) or not ids.is_cuda:
    return None
  pool = params.__dict__.setdefault("_plan_pool", {"free": [], "made": 0, "stream": None})
  if pool["free"]:
    plan = pool["free"].pop()
  elif pool["made"] < PLAN_POOL_MAX:
    plan = SparsePlan(params._primary, params.dim)
    pool["made"] += 1
  else:
    return None   # more lookups in flight than plans: this one takes the one-call write-back
  if pool["stream"] is None:
    pool["stream"] = torch.cuda.Stream(device=params._primary)
  side = pool["stream"]
  side.wait_stream(torch.cuda.current_stream(params._primary))   # the ids may still be being produced
  with torch.cuda.stream(side):
    plan.build(ids)
  return plan


def _release_plan(params, plan):
  pool = getattr(params, "_plan_pool", None)
  if pool is not None and plan is not None:
    pool["free"].append(plan)   # (its next build waits for the event of its last use: SparsePlan.build)


class TrainableWrapper:
  """The local [N,dim] "shadow" of the rows of one lookup (PY/embedding_weights.py:38-540):
  refilled from the table on read (`prefetch_values`), written back by `update_op`."""

  def __init__(self, params, ids, max_norm=None, plan_writeback=False):
    self.params = params
    self.ids = ids
    self.max_norm = max_norm
    self._values = None
    self.exists = None
    self.plan = _plan_at_lookup(params, ids) if plan_writeback else None
    self.prefetch_values()

  def take_plan(self):
    """The write-back plan started at lookup time (or None); the caller applies it once and hands it back to the pool."""
    plan, self.plan = self.plan, None
    return plan

  def __del__(self):
    try:
      if self.plan is not None:
        _release_plan(self.params, self.take_plan())
    except Exception:
      pass

  def prefetch_values(self):
    """PY/embedding_weights.py:163-170"""
    if self.params.bp_v2:
      r, self.exists = self.params.lookup(self.ids, return_exists=True)
    else:
      r = self.params.lookup(self.ids)
    self._values = self.transform(r)
    return self._values

  def transform(self, result):
    """PY/embedding_weights.py:497-521: optional clip_by_norm over the embedding axis."""
    if self.max_norm is None:
      return result
    norm = result.to(torch.float32).norm(dim=-1, keepdim=True)
    scale = self.max_norm / torch.maximum(norm, torch.full_like(norm, self.max_norm))
    return (result * scale).to(result.dtype)

  def read_value(self):
    return self._values

  def update_op(self, new_values, old_values=None):
    """PY/embedding_weights.py:434-444: upsert (or accum when bp_v2) the post-optimizer rows."""
    if self.params.bp_v2:
      old = self._values if old_values is None else old_values
      self.params.accum(self.ids, old, new_values, self.exists)
    else:
      self.params.upsert(self.ids, new_values)
    if self.params.restrict_policy is not None:  # PY/embedding_weights.py:441-442
      self.params.restrict_policy.apply_update(self.ids)
    self._values = new_values


def embedding_lookup(params, ids, partition_strategy=None, name=None, validate_indices=None, max_norm=None,
                     return_trainable=False, plan_writeback=False):
  """PY/dynamic_embedding_variable.py:1362-1530.  A miss returns the initializer row and does NOT
  insert (keys enter the table on the optimizer write-back).

  plan_writeback (with return_trainable): start the id-only half of the optimizer write-back of these ids now, on the
  Variable's second stream, so that it builds while the model's forward and backward run and `apply_gradients` finds it
  ready (no look-ahead into the next batch needed).  Opt-in: it pays when there is work between lookup and
  apply_gradients; back to back (bench shape, 131 072 ids) the extra Python — a stream context and three events — costs
  more than the overlap gains: 96 us per step against 75 us for lookup + one-call apply_sparse."""
  ids = torch.as_tensor(ids, device=params._primary)
  tw = TrainableWrapper(params, ids.reshape(-1), max_norm=max_norm, plan_writeback=return_trainable and plan_writeback)
  emb = tw.read_value().reshape(tuple(ids.shape) + (params.dim,))
  return (emb, tw) if return_trainable else emb


def embedding_lookup_unique(params, ids, partition_strategy=None, name=None, validate_indices=None, max_norm=None,
                            return_trainable=False):
  """PY/dynamic_embedding_ops.py:64-117: unique -> lookup -> gather."""
  ids = torch.as_tensor(ids, device=params._primary)
  uniq, idx, _ = device_ops.unique(ids)
  r = embedding_lookup(params, uniq, max_norm=max_norm, return_trainable=return_trainable)
  ue, tw = r if return_trainable else (r, None)
  emb = device_ops.gather_rows(ue, idx).reshape(tuple(ids.shape) + (params.dim,))
  return (emb, tw) if return_trainable else emb


def embedding_lookup_sparse(params, sp_ids, sp_weights=None, partition_strategy=None, name="embedding_lookup_sparse",
                            combiner="mean", max_norm=None, return_trainable=False, num_rows=None):
  """PY/dynamic_embedding_ops.py:120-293.  `sp_ids` = (indices[nnz,2] or row_ids[nnz], values[nnz]);
  `sp_weights` = matching weight values or None.  Segment combine sum / mean / sqrtn over rows."""
  if combiner not in ("mean", "sqrtn", "sum"):
    raise ValueError("combiner must be one of 'mean', 'sqrtn' or 'sum'")
  indices, ids = sp_ids
  indices = torch.as_tensor(indices, device=params._primary)
  seg = (indices[:, 0] if indices.dim() == 2 else indices).to(torch.int64)
  ids = torch.as_tensor(ids, device=params._primary)
  uniq, idx, _ = device_ops.unique(ids)
  r = embedding_lookup(params, uniq, max_norm=max_norm, return_trainable=return_trainable)
  ue, tw = r if return_trainable else (r, None)
  n = int(seg.max().item()) + 1 if num_rows is None else num_rows
  # gather + weights + segment combine fused in one kernel (reads the unique rows through idx)
  out = device_ops.sparse_segment_combine(ue, idx, seg, sp_weights if sp_weights is None else torch.as_tensor(
      sp_weights, dtype=torch.float32, device=ue.device), combiner, n)
  return (out, tw) if return_trainable else out


def safe_embedding_lookup_sparse(params, sp_ids, sparse_weights=None, combiner="mean", default_id=None,
                                 name="safe_embedding_lookup_sparse", partition_strategy=None, max_norm=None,
                                 return_trainable=False, num_rows=None):
  """PY/dynamic_embedding_ops.py:296-430.  `sp_ids` = (indices[nnz, R], values[nnz][, dense_shape[R]]) — a
  SparseTensor of rank R >= 2 (or row ids [nnz] for rank 2).  Semantics of the reference, NOT of
  `tf.nn.safe_embedding_lookup_sparse`: ids are never pruned (any int64 is a legal key, negative ones too,
  T/dynamic_embedding_ops_test.py:1007-1050); entries with weight <= 0 are dropped unless combiner == "sum"
  (`_prune_invalid_weights`, :374-376); rows left without entries yield zeros, or the embedding of
  `default_id` (`sparse_fill_empty_rows`, :379-408); leading dims are flattened for the lookup and restored
  on the result (:356-367, 411-424)."""
  if combiner not in ("mean", "sqrtn", "sum"):
    raise ValueError("combiner must be one of 'mean', 'sqrtn' or 'sum'")
  dense_shape = None
  if len(sp_ids) == 3:
    indices, ids, dense_shape = sp_ids
    dense_shape = [int(x) for x in dense_shape]
  else:
    indices, ids = sp_ids
  indices = torch.as_tensor(indices, device=params._primary)
  ids = torch.as_tensor(ids, device=params._primary)
  lead = None
  if indices.dim() == 2 and indices.shape[1] > 2:
    if dense_shape is None:
      raise ValueError("sparse ids of rank > 2 need their dense_shape: sp_ids = (indices, values, dense_shape)")
    lead = dense_shape[:-1]
    rows = torch.zeros(indices.shape[0], dtype=torch.int64, device=indices.device)
    for d in range(len(lead)):                       # row-major flattening of the leading dims
      rows = rows * lead[d] + indices[:, d].to(torch.int64)
    n = 1
    for d in lead:
      n *= d
  else:
    rows = (indices[:, 0] if indices.dim() == 2 else indices).to(torch.int64)
    if dense_shape is not None:
      n = dense_shape[0]
    elif num_rows is not None:
      n = num_rows
    else:
      n = int(rows.max().item()) + 1 if rows.numel() else 0
  w = None
  keep = None
  if sparse_weights is not None:
    w = torch.as_tensor(sparse_weights, dtype=torch.float32, device=params._primary)
    if combiner != "sum":
      keep = w > 0
  if keep is not None:
    rows, ids, w = rows[keep], ids[keep], w[keep]
  out = embedding_lookup_sparse(params, (rows, ids), w, combiner=combiner, max_norm=max_norm,
                                return_trainable=return_trainable, num_rows=n)
  res, tw = out if return_trainable else (out, None)
  if default_id is not None and n:
    empty = torch.ones(n, dtype=torch.bool, device=res.device)
    empty[rows] = False
    d = embedding_lookup(params, torch.tensor([default_id], dtype=torch.int64, device=params._primary),
                         max_norm=max_norm).to(torch.float32)
    res = torch.where(empty[:, None], d, res)
  if lead is not None:
    res = res.reshape(tuple(lead) + (res.shape[-1],))
  return (res, tw) if return_trainable else res
