"""KV-table wrappers: `CuckooHashTable` / `HkvHashTable` with the reference's method surface.

Mirrors PY/cuckoo_hashtable_ops.py:45-575 and PY/hkv_hashtable_ops.py:47-560
(PY = /root/reference/tensorflow_recommenders_addons/dynamic_embedding/python/ops): same
constructor arguments, same methods (`size, lookup, insert, accum, remove, clear, export,
save_to_file_system, load_from_file_system`; Hkv adds `export_with_scores,
export_keys_and_scores`), same argument meaning and error behaviour.  Tensors are torch
tensors on the table's MI355X; every method is one call through the C ABI
(include/tfra_mi355x.h) on torch's current HIP stream — no per-op host sync (the reference
syncs 1-3x per op, R/kernels/hkv_hashtable_op_gpu.cu.cc:192-213).

In the reference `CuckooHashTable` on a GPU device silently issues the Hkv ops
(PY/cuckoo_hashtable_ops.py:153-165,309-338); here both classes are the same HIP engine:
`CuckooHashTable` = unbounded, growing, never-evicting flavour (libcuckoo semantics),
`HkvHashTable` = bounded `max_capacity` flavour with scores.
"""
import ctypes
import enum
import os
import sys

import torch

from .. import _capi

_TORCH2DT = {
    torch.float32: _capi.TFRA_F32,
    torch.float16: _capi.TFRA_F16,
    torch.bfloat16: _capi.TFRA_BF16,
    torch.int8: _capi.TFRA_I8,
    torch.int32: _capi.TFRA_I32,
    torch.int64: _capi.TFRA_I64,
    torch.float64: _capi.TFRA_F64,
}

KHkvHashTableInitCapacity = 1024 * 1024  # PY/hkv_hashtable_ops.py:40-44
KHkvHashTableMaxCapacity = 1024 * 1024
KHkvHashTableMaxHbmForValuesByBytes = 1024 * 1024 * 1024


class HkvEvictStrategy(enum.IntEnum):
  """PY/dynamic_embedding_creator.py:141-146"""
  LRU = 0
  LFU = 1
  EPOCHLRU = 2
  EPOCHLFU = 3
  CUSTOMIZED = 4


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
  """The current HIP stream of `device` as a void*.  torch.cuda.current_stream() builds a Stream object through several
  Python layers (5 us per call, 18 calls in one routed multi-GPU step): the raw getter is the same value in 0.3 us."""
  if _raw_stream is not None:
    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    return ctypes.c_void_p(_raw_stream(idx if idx is not None else torch.cuda.current_device()))
  return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _as_device(device):
  if device is None or device == "" or device == []:
    device = "cuda:0"
  if isinstance(device, (list, tuple)):
    device = device[0]
  if isinstance(device, str):
    d = device.strip().lower().replace("/", "")
    # TF-style '/GPU:0' / '/device:GPU:0'
    if "gpu" in d:
      device = "cuda:" + d.split(":")[-1]
  dev = torch.device(device)
  if dev.type != "cuda":
    raise RuntimeError(
        "tfra_amd tables live on an MI355X (device %r requested); there is no CPU fallback" %
        (device,))
  if not torch.cuda.is_available():
    raise RuntimeError("no HIP device visible: tfra_amd has no CPU fallback")
  return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())


class _DeviceTable:
  """Owns one tfra_table_t and exposes the table ops on torch tensors."""

  def __init__(self, key_dtype, value_dtype, default_value, name, device, dim=None, aux_fields=0,
               init_capacity=0, max_capacity=0, max_hbm_for_values=0, strategy=-1, step_per_epoch=0,
               reserved_key_start_bit=0, max_load_factor=0.0, aux_init=(0.0, 0.0, 0.0, 0.0)):
    if key_dtype not in (torch.int64, torch.int32):
      # GPU ops: K = int64 (R/kernels/hkv_hashtable_op_gpu.cu.cc:1133-1138) and, for the cuckoo ops, (int32, float)
      # (R/kernels/cuckoo_hashtable_op_gpu.cu.cc:1058).  The engine's keys are int64: int32 keys are widened on the device in
      # front of every call (tfra_keys_widen_i32), exports narrow them, and the key files hold 4-byte keys like the reference's.
      raise TypeError("key_dtype must be torch.int64 or torch.int32 on GPU tables, got %s" % key_dtype)
    if value_dtype not in _TORCH2DT:
      raise TypeError("unsupported value_dtype %s" % value_dtype)
    self._key_dtype = key_dtype
    self._value_dtype = value_dtype
    self._device = _as_device(device)
    self._name = name
    dv = torch.as_tensor(default_value, dtype=value_dtype).reshape(-1)
    if dim is None:
      dim = dv.numel()
    elif dv.numel() == 1 and dim != 1:
      dv = dv.repeat(dim)
    if dv.numel() != dim:
      raise ValueError("default_value must be a vector of dim %d, got shape %s" % (dim, tuple(dv.shape)))
    self._dim = int(dim)
    self._default_value = dv.to(self._device).contiguous()
    o = _capi.TableOpts()
    o.struct_size = ctypes.sizeof(_capi.TableOpts)
    o.value_dtype = _TORCH2DT[value_dtype]
    o.dim = self._dim
    o.aux_fields = aux_fields
    o.init_capacity = int(init_capacity)
    o.max_capacity = int(max_capacity)
    o.max_hbm_for_vectors = int(max_hbm_for_values)
    o.max_load_factor = float(max_load_factor)
    o.strategy = int(strategy)
    o.step_per_epoch = int(step_per_epoch)
    o.reserved_key_start_bit = int(reserved_key_start_bit)
    o.device = self._device.index
    for i in range(4):
      o.aux_init[i] = float(aux_init[i]) if i < len(aux_init) else 0.0
    self._aux_fields = aux_fields
    h = ctypes.c_void_p()
    _capi.call("tfra_table_create", ctypes.byref(o), None, ctypes.byref(h))
    self._h = h
    if key_dtype == torch.int32:
      _capi.call("tfra_table_set_option", self._h, _capi.OPTION_KEY_BYTES_ON_DISK, 4)

  def __del__(self):
    h = getattr(self, "_h", None)
    if h:
      try:
        _capi.lib().tfra_table_destroy(h)
      except Exception:  # interpreter shutdown
        pass
      self._h = None

  # ---- helpers -----------------------------------------------------------------------------
  @property
  def dim(self):
    return self._dim

  @property
  def device(self):
    return self._device

  @property
  def key_dtype(self):
    return self._key_dtype

  @property
  def value_dtype(self):
    return self._value_dtype

  def _keys(self, keys):
    keys = torch.as_tensor(keys, device=self._device) if not torch.is_tensor(keys) else keys
    if keys.dtype != self._key_dtype:
      raise TypeError("Signature mismatch. Keys must be dtype %s, got %s." % (self._key_dtype, keys.dtype))
    keys = keys.to(self._device).contiguous()
    if self._key_dtype == torch.int32:   # the engine's keys are int64
      wide = torch.empty(keys.shape, dtype=torch.int64, device=self._device)
      _capi.call("tfra_keys_widen_i32", keys.numel(), _ptr(keys), _ptr(wide), _stream(self._device))
      return wide
    return keys

  def _values_for(self, keys, values, what="values"):
    values = torch.as_tensor(values, device=self._device) if not torch.is_tensor(values) else values
    if values.dtype != self._value_dtype:
      raise TypeError("Signature mismatch. %s must be dtype %s, got %s." % (what, self._value_dtype, values.dtype))
    want = tuple(keys.shape) + (self._dim,)
    if tuple(values.shape) != want:
      # CheckKeyAndValueTensorsForInsert (R/kernels/cuckoo_hashtable_op.cc:640-660), KAT K6
      raise ValueError("Expected shape %s for %s, got %s" % (list(want), what, list(values.shape)))
    return values.to(self._device).contiguous()

  # ---- ops ---------------------------------------------------------------------------------
  def find(self, keys, dynamic_default_values=None, return_exists=False, field=0):
    keys = self._keys(keys)
    n = keys.numel()
    d = self._default_value if dynamic_default_values is None else dynamic_default_values
    d = torch.as_tensor(d, device=self._device) if not torch.is_tensor(d) else d.to(self._device)
    if d.dtype != self._value_dtype:
      raise TypeError("default values must be dtype %s, got %s" % (self._value_dtype, d.dtype))
    d = d.contiguous()
    out = torch.empty(tuple(keys.shape) + (self._dim,), dtype=self._value_dtype, device=self._device)
    # is_full_default = (value_flat.size() == default_flat.size())
    # (R/kernels/hkv_hashtable_op_gpu.cu.cc:188-190, cuckoo_hashtable_op.cc:48-50)
    full = int(out.numel() == d.numel())
    if not full and d.numel() < self._dim:
      raise ValueError("default value needs at least dim=%d elements, got %d" % (self._dim, d.numel()))
    exists = torch.empty(keys.shape, dtype=torch.bool, device=self._device) if return_exists else None
    if n:
      if field:
        _capi.call("tfra_table_find_field", self._h, field, n, _ptr(keys), _ptr(out), _ptr(exists), _ptr(d), full,
                   _stream(self._device))
      else:
        _capi.call("tfra_table_find", self._h, n, _ptr(keys), _ptr(out), _ptr(exists), _ptr(d), full,
                   _stream(self._device))
    return (out, exists) if return_exists else out

  def find_unique(self, keys, dynamic_default_values=None, return_exists=False):
    """find(keys) and the de-duplication of the same keys in ONE launch (tfra_table_find_unique: the forward half of embedding_lookup
    as the fused TF op issues it): returns (rows [n, dim], unique [n] — the first `count` entries are the distinct keys, in no
    particular order —, idx [n] int32 with unique[idx] == keys, count: device int64 scalar[, exists]).  Nothing is read on the host."""
    from .device_ops import _workspace
    keys = self._keys(keys).reshape(-1)
    n = keys.numel()
    d = self._default_value if dynamic_default_values is None else dynamic_default_values
    d = torch.as_tensor(d, device=self._device) if not torch.is_tensor(d) else d.to(self._device)
    if d.dtype != self._value_dtype:
      raise TypeError("default values must be dtype %s, got %s" % (self._value_dtype, d.dtype))
    d = d.contiguous()
    out = torch.empty((n, self._dim), dtype=self._value_dtype, device=self._device)
    full = int(out.numel() == d.numel())
    if not full and d.numel() < self._dim:
      raise ValueError("default value needs at least dim=%d elements, got %d" % (self._dim, d.numel()))
    exists = torch.empty(n, dtype=torch.bool, device=self._device) if return_exists else None
    uniq = torch.empty(n, dtype=torch.int64, device=self._device)
    idx = torch.empty(n, dtype=torch.int32, device=self._device)
    cnt = torch.zeros((), dtype=torch.int64, device=self._device)
    _capi.call("tfra_table_find_unique", self._h, _workspace(self._device), n, _ptr(keys), _ptr(out), _ptr(exists), _ptr(d), full,
               _ptr(uniq), _ptr(idx), _ptr(cnt), _stream(self._device))
    return (out, uniq, idx, cnt, exists) if return_exists else (out, uniq, idx, cnt)

  def upsert(self, keys, values, scores=None, unique_keys=False, field=0):
    keys = self._keys(keys)
    values = self._values_for(keys, values)
    n = keys.numel()
    if n == 0:
      return
    flags = _capi.FLAG_UNIQUE_KEYS if unique_keys else 0
    if scores is not None and scores.numel() == 0:
      scores = None  # HkvHashTableInsert: empty scores tensor == no scores
    if scores is not None:
      scores = scores.to(self._device, torch.int64).contiguous()
      if scores.numel() != n:
        raise ValueError("scores must have one entry per key")
    if field:
      _capi.call("tfra_table_insert_field", self._h, field, n, _ptr(keys), _ptr(values), flags, _stream(self._device))
    else:
      _capi.call("tfra_table_insert_or_assign", self._h, n, _ptr(keys), _ptr(values), _ptr(scores), flags,
                 _stream(self._device))

  def find_n(self, keys, count, out=None, return_exists=False):
    """find over the first `count[0]` entries of the buffer `keys`, the count read ON THE DEVICE (`count`: an int64 tensor on the
    table's device, or pinned host memory): tfra_table_find_n.  Rows beyond the count are left as they are."""
    keys = self._keys(keys)
    n = keys.numel()
    d = self._default_value.contiguous()
    if out is None:
      out = torch.empty((n, self._dim), dtype=self._value_dtype, device=self._device)
    exists = torch.zeros(n, dtype=torch.bool, device=self._device) if return_exists else None
    if n:
      _capi.call("tfra_table_find_n", self._h, n, _ptr(count), _ptr(keys), _ptr(out), _ptr(exists), _ptr(d), 0, _stream(self._device))
    return (out, exists) if return_exists else out

  def upsert_n(self, keys, count, values, scores=None):
    """insert_or_assign of the first `count[0]` UNIQUE keys of the buffer, the count read on the device
    (tfra_table_insert_or_assign_n; raises when the single-pass write-back cannot take the call)."""
    keys = self._keys(keys)
    values = self._values_for(keys, values)
    if scores is not None:
      scores = scores.to(self._device, torch.int64).contiguous()
    if keys.numel():
      _capi.call("tfra_table_insert_or_assign_n", self._h, keys.numel(), _ptr(count), _ptr(keys), _ptr(values), _ptr(scores),
                 _stream(self._device))

  def accum_or_assign(self, keys, values_or_deltas, exists, scores=None, unique_keys=False):
    keys = self._keys(keys)
    vod = self._values_for(keys, values_or_deltas, "values_or_deltas")
    exists = torch.as_tensor(exists, device=self._device)
    if exists.dtype != torch.bool:
      raise TypeError("exists must be a bool tensor")
    if exists.numel() != keys.numel():
      raise ValueError("exists must have the same number of elements as keys")
    exists = exists.contiguous()
    n = keys.numel()
    if n == 0:
      return
    if scores is not None and scores.numel() == 0:
      scores = None
    if scores is not None:
      scores = scores.to(self._device, torch.int64).contiguous()
    _capi.call("tfra_table_accum_or_assign", self._h, n, _ptr(keys), _ptr(vod), _ptr(exists), _ptr(scores),
               _capi.FLAG_UNIQUE_KEYS if unique_keys else 0, _stream(self._device))

  def erase(self, keys):
    keys = self._keys(keys)
    if keys.numel():
      _capi.call("tfra_table_erase", self._h, keys.numel(), _ptr(keys), _stream(self._device))

  def clear_all(self):
    _capi.call("tfra_table_clear", self._h, _stream(self._device))

  def size_host(self):
    out = ctypes.c_size_t()
    _capi.call("tfra_table_size", self._h, ctypes.byref(out), _stream(self._device))
    return out.value

  def size_device(self):
    out = torch.empty((), dtype=torch.int64, device=self._device)
    _capi.call("tfra_table_size_to_device", self._h, _ptr(out), _stream(self._device))
    return out

  def capacity(self):
    out = ctypes.c_size_t()
    _capi.call("tfra_table_capacity", self._h, ctypes.byref(out))
    return out.value

  def check_errors(self):
    """Raises TfraError if the device dropped keys since the last check (table full, plan overflow); synchronises."""
    _capi.call("tfra_table_check_errors", self._h, _stream(self.device))

  def slot_census(self):
    """{empty, locked, live, ovf0, ovf1}: key-slot and bucket-flag counts (introspection; synchronises)."""
    out = (ctypes.c_uint64 * 5)()
    _capi.call("tfra_table_slot_census", self._h, out, _stream(self.device))
    return dict(zip(("empty", "locked", "live", "ovf0", "ovf1"), [int(x) for x in out]))

  def growth_stats(self):
    """{growths, in_place, mapped_range, mapped_bytes} (tfra_table_growth_stats)."""
    out = (ctypes.c_uint64 * 4)()
    _capi.call("tfra_table_growth_stats", self._h, out)
    return dict(zip(("growths", "in_place", "mapped_range", "mapped_bytes"), [int(x) for x in out]))

  def set_capture_safe(self, on):
    """While True every op on this table can be captured into a HIP graph (no host sync, no growth)."""
    _capi.call("tfra_table_set_option", self._h, _capi.OPTION_CAPTURE_SAFE, int(bool(on)))

  def set_owner_tags(self, on):
    """False: planned write-backs take the general two-kernel path (what runs when the tag array cannot be allocated)."""
    _capi.call("tfra_table_set_option", self._h, _capi.OPTION_NO_OWNER_TAGS, int(not on))

  def reserve(self, n_slots):
    _capi.call("tfra_table_reserve", self._h, int(n_slots), _stream(self._device))

  def export_all(self, with_scores=False, values=True, split_size=None):
    """Export = size, then export_batch over the capacity (hkv_hashtable_op_gpu.cu.cc:425-470)."""
    n = self.size_host()
    cap = self.capacity()
    keys = torch.empty(n, dtype=torch.int64, device=self._device)
    vals = torch.empty((n, self._dim), dtype=self._value_dtype, device=self._device) if values else None
    scores = torch.empty(n, dtype=torch.int64, device=self._device) if with_scores else None
    counter = torch.zeros(1, dtype=torch.int64, device=self._device)
    step = cap if not split_size else int(split_size)
    for off in range(0, cap, step):
      _capi.call("tfra_table_export_batch", self._h, min(step, cap - off), off, _ptr(counter), _ptr(keys), _ptr(vals),
                 _ptr(scores), _stream(self._device))
    got = int(counter.item())
    if got != n:
      raise RuntimeError("export: table changed during export (%d vs %d)" % (got, n))
    if self._key_dtype == torch.int32:
      narrow = torch.empty(n, dtype=torch.int32, device=self._device)
      _capi.call("tfra_keys_narrow_i32", n, _ptr(keys), _ptr(narrow), None, _stream(self._device))
      keys = narrow
    return keys, vals, scores

  def save(self, prefix, buffer_size=4194304, append_to_file=False, field=0):
    """field > 0: the co-located state vector `field` (an optimizer slot) in the same file format."""
    self.check_errors()   # a sync point anyway: insert / apply failures must not pass silently into a checkpoint
    out = ctypes.c_size_t()
    _capi.call("tfra_table_save_field", self._h, int(field), prefix.encode(), int(buffer_size), int(bool(append_to_file)),
               _stream(self._device), ctypes.byref(out))
    return out.value

  def load(self, prefix, buffer_size=4194304, field=0):
    out = ctypes.c_size_t()
    _capi.call("tfra_table_load_field", self._h, int(field), prefix.encode(), int(buffer_size), _stream(self._device),
               ctypes.byref(out))
    return out.value

  def apply_optimizer(self, params, keys, grads, param_defaults, n_dev=None):
    """n_dev: optional device int64 scalar = number of leading (key, grad) pairs that are valid."""
    keys = self._keys(keys)
    grads = grads.to(self._device, torch.float32).contiguous()
    if tuple(grads.shape) != tuple(keys.shape) + (self._dim,):
      raise ValueError("Expected shape %s for grads, got %s" % (list(keys.shape) + [self._dim], list(grads.shape)))
    d = param_defaults.to(self._device, torch.float32).contiguous()
    full = int(d.numel() == grads.numel())
    _capi.call("tfra_table_apply_optimizer", self._h, ctypes.byref(params), keys.numel(), _ptr(keys), _ptr(grads), _ptr(d),
               full, _ptr(n_dev), _stream(self._device))


  # method of _DeviceTable (appended below the class body by assignment)
def _apply_sparse(self, params, ids, grads, default_row):
  """ids may repeat: duplicate gradients are summed (fixed order), one fused update per unique key."""
  ids = self._keys(ids).reshape(-1)
  grads = grads.to(self._device, torch.float32).contiguous()
  if grads.numel() != ids.numel() * self._dim:
    raise ValueError("Expected shape %s for grads, got %s" % ([ids.numel(), self._dim], list(grads.shape)))
  d = default_row.to(self._device, torch.float32).contiguous()
  _capi.call("tfra_table_apply_sparse", self._h, ctypes.byref(params), ids.numel(), _ptr(ids), _ptr(grads), _ptr(d),
             _stream(self._device))


_DeviceTable.apply_sparse = _apply_sparse


class SparsePlan:
  """The id-only half of `apply_sparse` for one batch (tfra_sparse_plan_*): which ids repeat, the order
  their gradients are summed in, the unique keys of the step.  Build it as soon as the ids are known —
  on a side stream next to the lookup of the same ids, or while the previous step is still running —
  and hand it to `_DeviceTable.apply_planned` with the gradients.  Stream ordering is handled here:
  `apply_planned` waits for the build, a rebuild waits for the last `apply_planned` that used the plan."""

  def __init__(self, device, dim):
    self._device = _as_device(device)
    self._dim = int(dim)
    self._h = ctypes.c_void_p()
    _capi.call("tfra_sparse_plan_create", self._device.index, ctypes.byref(self._h))
    self._built = torch.cuda.Event()
    self._used = None
    self.ids = None
    self.n = 0

  def build(self, ids, sync=True):
    """Enqueue the build on the CURRENT stream of the plan's device.  sync=False leaves the ordering
    against `apply_planned` to the caller (a captured HIP graph orders them by its edges)."""
    ids = ids.to(self._device, torch.int64).contiguous().reshape(-1)
    stream = torch.cuda.current_stream(self._device)
    if sync:
      if self._used is not None:
        stream.wait_event(self._used)     # the previous batch's sums/apply still read the plan buffers
      ids.record_stream(stream)
    self.ids, self.n = ids, ids.numel()   # keeps the ids alive until the next build
    _capi.call("tfra_sparse_plan_build", self._h, self.n, _ptr(ids), self._dim, _stream(self._device))
    if sync:
      self._built.record(stream)
    return self

  def partition(self, num_shards, mode=0):
    """The plan's distinct ids grouped by owner (tfra_plan_partition: tfra_partition over the plan's keys, in place of
    tf.unique + dynamic_partition): owner-major ids [n], perm [n] (plan index of each owner-major row), counts
    [num_shards] (device).  Entries past sum(counts) are unspecified.  Runs on the current stream after the build."""
    from .device_ops import _workspace
    dev = self._device
    keys_out = torch.empty(self.n, dtype=torch.int64, device=dev)
    perm = torch.empty(self.n, dtype=torch.int32, device=dev)
    counts = torch.zeros(num_shards, dtype=torch.int64, device=dev)
    torch.cuda.current_stream(dev).wait_event(self._built)
    _capi.call("tfra_plan_partition", self._h, _workspace(dev), int(num_shards), int(mode), _ptr(keys_out), _ptr(perm), _ptr(counts),
               _stream(dev))
    return keys_out, perm, counts

  def positions_to(self, perm):
    """dest[p] = j for every batch position p whose id is the key of owner-major row j (tfra_plan_positions_to)."""
    dest = torch.empty(self.n, dtype=torch.int32, device=self._device)
    _capi.call("tfra_plan_positions_to", self._h, _ptr(perm), _ptr(dest), _stream(self._device))
    return dest

  def reduce_to(self, grads, dest, rows_out, sync=True):
    """rows_out[dest[p], :] = sum of the gradient rows of the id at position p (tfra_plan_reduce_to): the per-key sums of
    the plan's batch, scattered by a caller-supplied position -> row map (equal for all positions of an id)."""
    grads = grads.to(self._device, torch.float32).contiguous()
    if grads.numel() != self.n * self._dim:
      raise ValueError("Expected shape %s for grads, got %s" % ([self.n, self._dim], list(grads.shape)))
    if dest.dtype != torch.int32 or dest.numel() != self.n or not dest.is_contiguous():
      raise ValueError("dest must be a contiguous int32 tensor with one entry per id")
    stream = torch.cuda.current_stream(self._device)
    if sync:
      stream.wait_event(self._built)
    _capi.call("tfra_plan_reduce_to", self._h, _ptr(grads), _ptr(dest), _ptr(rows_out), _stream(self._device))
    if sync:
      if self._used is None:
        self._used = torch.cuda.Event()
      self._used.record(stream)
    return rows_out

  def read(self):
    """The batch as CSR-by-key, on the host (tests / tools): counts dict, keys [U], cnt [U], positions [n]
    (the positions of key 0, of key 1, ..., each ascending).  Synchronises the current stream."""
    import numpy as np
    counts = (ctypes.c_uint32 * 6)()
    _capi.call("tfra_sparse_plan_read", self._h, counts, None, None, None, 0, _stream(self._device))
    u = counts[0] + counts[1]
    keys = np.empty(u, np.int64)
    cnt = np.empty(u, np.uint32)
    pos = np.empty(self.n, np.uint32) if self._dim else None   # an assign-only plan (dim 0) keeps the last position of a key only
    _capi.call("tfra_sparse_plan_read", self._h, counts, keys.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p),
               pos.ctypes.data_as(ctypes.c_void_p) if pos is not None else None, u, _stream(self._device))
    names = ("many", "few", "partials", "bins", "few_entries", "errors")
    return dict(zip(names, list(counts))), keys, cnt, pos

  def __del__(self):
    try:
      if self._h:
        _capi.call("tfra_sparse_plan_destroy", self._h)
        self._h = None
    except Exception:
      pass


def _apply_planned(self, params, plan, grads, default_row, sync=True):
  """Gradient half of apply_sparse for a batch whose id-only half was built ahead (`SparsePlan`)."""
  if plan._dim != self._dim or plan._device != self._device:
    raise ValueError("the plan was built for dim %d on %s" % (plan._dim, plan._device))
  grads = grads.to(self._device, torch.float32).contiguous()
  if grads.numel() != plan.n * self._dim:
    raise ValueError("Expected shape %s for grads, got %s" % ([plan.n, self._dim], list(grads.shape)))
  d = default_row.to(self._device, torch.float32).contiguous()
  stream = torch.cuda.current_stream(self._device)
  if sync:
    stream.wait_event(plan._built)
  _capi.call("tfra_table_apply_planned", self._h, ctypes.byref(params), plan._h, _ptr(grads), _ptr(d), _stream(self._device))
  if sync:
    if plan._used is None:
      plan._used = torch.cuda.Event()
    plan._used.record(stream)


_DeviceTable.apply_planned = _apply_planned


def _upsert_sparse(self, ids, values, scores=None):
  """insert_or_assign of a batch whose keys may repeat: the LAST occurrence wins (the reference's sequential
  order), de-duplicated on the device — also on a bounded table at max_capacity."""
  ids = self._keys(ids).reshape(-1)
  values = self._values_for(ids, values.reshape(ids.numel(), -1))
  if scores is not None:
    scores = scores.to(self._device, torch.int64).contiguous()
  if ids.numel():
    _capi.call("tfra_table_upsert_sparse", self._h, ids.numel(), _ptr(ids), _ptr(values), _ptr(scores), _stream(self._device))


def _upsert_planned(self, plan, values, scores=None, sync=True):
  """Assign half of upsert_sparse for a batch whose id-only half was built ahead (`SparsePlan`)."""
  if plan._device != self._device:
    raise ValueError("the plan lives on %s" % (plan._device,))
  values = values.to(self._device).contiguous()
  if values.dtype != self._value_dtype or values.numel() != plan.n * self._dim:
    raise ValueError("Expected %s values of shape %s" % (self._value_dtype, [plan.n, self._dim]))
  if scores is not None:
    scores = scores.to(self._device, torch.int64).contiguous()
  stream = torch.cuda.current_stream(self._device)
  if sync:
    stream.wait_event(plan._built)
  _capi.call("tfra_table_upsert_planned", self._h, plan._h, _ptr(values), _ptr(scores), _stream(self._device))
  if sync:
    if plan._used is None:
      plan._used = torch.cuda.Event()
    plan._used.record(stream)


_DeviceTable.upsert_sparse = _upsert_sparse
_DeviceTable.upsert_planned = _upsert_planned


class _LookupInterfaceMirror:
  """Shared method surface of CuckooHashTable / HkvHashTable (tf LookupInterface subclasses)."""

  _table: _DeviceTable

  @property
  def name(self):
    return self._name

  @property
  def key_dtype(self):
    return self._key_dtype

  @property
  def value_dtype(self):
    return self._value_dtype

  @property
  def resource_handle(self):
    return self._table

  def size(self, name=None):
    """Scalar int64 DEVICE tensor (GPU `size_i64`, hkv_hashtable_op_gpu.cu.cc:172-179)."""
    return self._table.size_device()

  def remove(self, keys, name=None):
    """PY/cuckoo_hashtable_ops.py:219-246: absent keys are silently ignored."""
    self._table.erase(keys)

  def clear(self, name=None):
    self._table.clear_all()

  def lookup(self, keys, dynamic_default_values=None, return_exists=False, name=None):
    """PY/cuckoo_hashtable_ops.py:272-340"""
    return self._table.find(keys, dynamic_default_values, return_exists)

  def export(self, name=None):
    k, v, _ = self._table.export_all()
    return k, v

  def _file_prefix(self, dirpath, file_name, dirpath_env):
    # PY/cuckoo_hashtable_ops.py:437-480: env var wins over dirpath; file name defaults to table name
    if dirpath_env:
      dirpath = os.environ.get(dirpath_env, dirpath)
    if not dirpath:
      raise ValueError("dirpath is required")
    os.makedirs(dirpath, exist_ok=True)
    return os.path.join(dirpath, file_name if file_name else self._name)

  def save_to_file_system(self, dirpath, file_name=None, dirpath_env="TFRA_SAVED_KV", append_to_file=False,
                          buffer_size=4194304, name=None):
    return self._table.save(self._file_prefix(dirpath, file_name, dirpath_env), buffer_size, append_to_file)

  def load_from_file_system(self, dirpath, file_name=None, dirpath_env="TFRA_SAVED_KV", load_entire_dir=False,
                            buffer_size=4194304, name=None):
    """load_entire_dir: load every `*-keys` file of the directory (PY/cuckoo_hashtable_ops.py:482-523).
    The GPU op clears the table first (hkv_hashtable_op_gpu.cu.cc:619)."""
    prefix = self._file_prefix(dirpath, file_name, dirpath_env)
    self._table.clear_all()
    if load_entire_dir:
      d = os.path.dirname(prefix)
      total = 0
      for f in sorted(os.listdir(d)):
        if f.endswith("-keys"):
          total += self._table.load(os.path.join(d, f[:-len("-keys")]), buffer_size)
      return total
    return self._table.load(prefix, buffer_size)


class CuckooHashTable(_LookupInterfaceMirror):
  """PY/cuckoo_hashtable_ops.py:45-145.  Growing, never-evicting table (libcuckoo semantics)."""

  def __init__(self, key_dtype, value_dtype, default_value, name="CuckooHashTable", checkpoint=True, init_size=0,
               config=None, device="", shard_saveable_object_fn=None, dim=None, aux_fields=0, aux_init=(0.0,) * 4):
    self._key_dtype = key_dtype
    self._value_dtype = value_dtype
    self._name = name
    self._checkpoint = checkpoint
    self._init_size = init_size
    self._max_capacity = sys.maxsize
    if init_size == 0:
      # K/cuckoo_hashtable_op.cc:199-207: TF_HASHTABLE_INIT_SIZE, default 8192
      init_size = int(os.environ.get("TF_HASHTABLE_INIT_SIZE", 8192))
    self._table = _DeviceTable(key_dtype, value_dtype, default_value, name, device, dim=dim, aux_fields=aux_fields,
                               init_capacity=init_size, max_capacity=0, strategy=-1, aux_init=aux_init)
    self._default_value = self._table._default_value
    self._device = self._table.device

  def insert(self, keys, values, name=None):
    """PY/cuckoo_hashtable_ops.py:342-373.  Duplicate keys: last one wins (sequential CPU order)."""
    self._table.upsert(keys, values)

  def accum(self, keys, values_or_deltas, exists, name=None):
    """PY/cuckoo_hashtable_ops.py:375-412"""
    self._table.accum_or_assign(keys, values_or_deltas, exists)


class HkvHashTable(_LookupInterfaceMirror):
  """PY/hkv_hashtable_ops.py:47-175.  Bounded table with per-key scores and in-bucket eviction."""

  def __init__(self, key_dtype, value_dtype, default_value, name="HkvHashTable", checkpoint=True,
               init_capacity=KHkvHashTableInitCapacity, max_capacity=KHkvHashTableMaxCapacity,
               max_hbm_for_values=KHkvHashTableMaxHbmForValuesByBytes, config=None, device="",
               shard_saveable_object_fn=None, evict_strategy=HkvEvictStrategy.LRU, step_per_epoch=0, gen_scores_fn=None,
               reserved_key_start_bit=0, dim=None, aux_fields=0, aux_init=(0.0,) * 4):
    if config:
      init_capacity = config.init_capacity
      max_capacity = config.max_capacity
      max_hbm_for_values = config.max_hbm_for_values
      evict_strategy = config.evict_strategy
      step_per_epoch = config.step_per_epoch
      gen_scores_fn = config.gen_scores_fn
      reserved_key_start_bit = config.reserved_key_start_bit
    self._key_dtype = key_dtype
    self._value_dtype = value_dtype
    self._scores_dtype = torch.int64
    self._name = name
    self._checkpoint = checkpoint
    self._init_capacity = init_capacity
    self._max_capacity = max_capacity
    self._max_hbm_for_values = max_hbm_for_values
    self._evict_strategy = HkvEvictStrategy(evict_strategy)
    self._step_per_epoch = step_per_epoch
    self._gen_scores_fn = gen_scores_fn
    self._reserved_key_start_bit = reserved_key_start_bit
    if max_capacity == 0:
      # hkv_hashtable_op_gpu.cu.cc:104-120
      env = os.environ.get("TFRA_GPU_HASHTABLE_UPLIMIT_SIZE")
      if env is None:
        raise ValueError("max_capaicty=0 and TFRA_GPU_HASHTABLE_UPLIMIT_SIZE not set is not valid.")
      max_capacity = int(env)
    self._table = _DeviceTable(key_dtype, value_dtype, default_value, name, device, dim=dim, aux_fields=aux_fields,
                               init_capacity=init_capacity, max_capacity=max_capacity,
                               max_hbm_for_values=max_hbm_for_values, strategy=int(self._evict_strategy),
                               step_per_epoch=step_per_epoch, reserved_key_start_bit=reserved_key_start_bit,
                               aux_init=aux_init)
    self._default_value = self._table._default_value
    self._device = self._table.device

  def _gen_scores(self, keys):
    """PY/hkv_hashtable_ops.py:209-217"""
    if self._evict_strategy == HkvEvictStrategy.CUSTOMIZED:
      assert self._gen_scores_fn is not None, "You must set gen_scores_fn when set evict strategy to CUSTOMIZED"
      return self._gen_scores_fn(keys)
    if self._evict_strategy in (HkvEvictStrategy.LFU, HkvEvictStrategy.EPOCHLFU):
      return torch.ones(keys.shape, dtype=torch.int64, device=keys.device)
    return None

  def insert(self, keys, values, name=None):
    """PY/hkv_hashtable_ops.py:339-367"""
    keys = self._table._keys(keys)
    # HKV requires unique keys per call (PY/dynamic_embedding_variable.py:1377-1378); with that contract
    # a full bounded table can evict by score
    self._table.upsert(keys, values, scores=self._gen_scores(keys), unique_keys=True)

  def accum(self, keys, values_or_deltas, exists, name=None):
    """PY/hkv_hashtable_ops.py:369-402"""
    keys = self._table._keys(keys)
    self._table.accum_or_assign(keys, values_or_deltas, exists, scores=self._gen_scores(keys), unique_keys=True)

  def export_keys_and_scores(self, split_size, name=None):
    """PY/hkv_hashtable_ops.py:421-434"""
    if not (isinstance(split_size, int) and split_size > 0):
      raise ValueError("split_size must be positive integer.")
    k, _, s = self._table.export_all(with_scores=True, values=False, split_size=split_size)
    return k, s

  def export_with_scores(self, split_size, name=None):
    """PY/hkv_hashtable_ops.py:436-450"""
    if not (isinstance(split_size, int) and split_size > 0):
      raise ValueError("split_size must be positive integer.")
    return self._table.export_all(with_scores=True, values=True, split_size=split_size)
