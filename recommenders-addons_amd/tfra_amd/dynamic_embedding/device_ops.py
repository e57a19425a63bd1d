"""Device front-end ops that bracket the table calls (N1/N3 of SURVEY.md §8f), as torch-tensor
functions over the C ABI: unique, gather, segment_sum, partition (+stitch)."""
import ctypes

import torch

from .. import _capi
from .table_ops import _ptr, _stream

_WS = {}


def _workspace(device):
  """One scratch workspace per (device, CURRENT STREAM): a workspace holds state that persists from call to call (tfra_unique's
  two self-emptying hash sets, their parity and generation), so two streams must never share one — calls on different
  streams are not ordered with each other."""
  key = (device.index, _stream(device).value)
  if key not in _WS:
    h = ctypes.c_void_p()
    _capi.call("tfra_workspace_create", device.index, ctypes.byref(h))
    _WS[key] = h
  return _WS[key]


def unique(ids, ordered=True):
  """tf.unique (first-occurrence order): returns (unique[U], idx[n] int32, num_unique device scalar).

  `unique` is returned as a length-n buffer view trimmed with ONE host read of the count — the
  same sync TF's `tf.unique` output-shape inference imposes (PY/dynamic_embedding_ops.py:99).
  ordered=False: the distinct ids in no particular order (tfra_unique_unordered, two launches instead of three; up to 2^18 ids)
  — all embedding_lookup needs: the order of tf.unique's output is not observable behind the gather."""
  ids = ids.contiguous()
  flat = ids.reshape(-1)
  n = flat.numel()
  dev = flat.device
  uniq = torch.empty(n, dtype=torch.int64, device=dev)
  idx = torch.empty(n, dtype=torch.int32, device=dev)
  cnt = torch.zeros((), dtype=torch.int64, device=dev)
  fn = "tfra_unique" if (ordered or n > (1 << 18)) else "tfra_unique_unordered"
  _capi.call(fn, _workspace(dev), n, _ptr(flat), _ptr(uniq), _ptr(idx), _ptr(cnt), _stream(dev))
  u = int(cnt.item())
  return uniq[:u], idx, cnt


def unique_no_sync(ids):
  """Like `unique` but never reads the count on the host: returns the full-length buffer."""
  flat = ids.contiguous().reshape(-1)
  n = flat.numel()
  dev = flat.device
  uniq = torch.empty(n, dtype=torch.int64, device=dev)
  idx = torch.empty(n, dtype=torch.int32, device=dev)
  cnt = torch.zeros((), dtype=torch.int64, device=dev)
  _capi.call("tfra_unique", _workspace(dev), n, _ptr(flat), _ptr(uniq), _ptr(idx), _ptr(cnt), _stream(dev))
  return uniq, idx, cnt


def gather_rows(rows, idx):
  """out[i,:] = rows[idx[i],:]  (tf.gather after unique, PY/dynamic_embedding_ops.py:111)."""
  rows = rows.contiguous()
  idx = idx.contiguous()
  n = idx.numel()
  row_bytes = rows.shape[-1] * rows.element_size()
  out = torch.empty((n, rows.shape[-1]), dtype=rows.dtype, device=rows.device)
  _capi.call("tfra_gather_rows", n, row_bytes, _ptr(rows), _ptr(idx), _ptr(out), _stream(rows.device))
  return out


def scatter_rows(rows, perm, n_out=None):
  """out[perm[i],:] = rows[i,:]  (dynamic_stitch with one flat permutation)."""
  rows = rows.contiguous()
  perm = perm.contiguous()
  n = perm.numel()
  two_d = rows.reshape(n, -1)
  row_bytes = two_d.shape[1] * rows.element_size()
  out = torch.empty((n if n_out is None else n_out, two_d.shape[1]), dtype=rows.dtype, device=rows.device)
  _capi.call("tfra_scatter_rows", n, row_bytes, _ptr(two_d), _ptr(perm), _ptr(out), _stream(rows.device))
  return out


def segment_sum(grads, idx, num_segments_dev, max_segments):
  """unsorted_segment_sum(grads, idx) -> [max_segments, dim]; rows >= *num_segments_dev are unspecified.
  Members of a segment are added in input order, one fp32 add each (sequential CPU order)."""
  grads = grads.contiguous()
  idx = idx.contiguous()
  n = idx.numel()
  dim = grads.shape[-1]
  out = torch.empty((max_segments, dim), dtype=torch.float32, device=grads.device)
  _capi.call("tfra_segment_sum", _workspace(grads.device), n, dim, _ptr(grads), _ptr(idx), _ptr(num_segments_dev),
             max_segments, _ptr(out), _stream(grads.device))
  return out


def reduce_by_key(ids, rows):
  """unique + unsorted_segment_sum, parallel per key (hot ids do not serialise) and bit-reproducible.
  Returns (keys[n] buffer, sums[n,dim] buffer, count device scalar): the first `count` entries are valid,
  in a deterministic but unspecified key order."""
  ids = ids.contiguous().reshape(-1)
  rows = rows.to(torch.float32).contiguous()
  n, dim = ids.numel(), rows.shape[-1]
  dev = ids.device
  keys = torch.empty(n, dtype=torch.int64, device=dev)
  sums = torch.empty((n, dim), dtype=torch.float32, device=dev)
  cnt = torch.zeros((), dtype=torch.int64, device=dev)
  _capi.call("tfra_reduce_by_key", _workspace(dev), n, _ptr(ids), dim, _ptr(rows), _ptr(keys), _ptr(sums), _ptr(cnt),
             _stream(dev))
  return keys, sums, cnt


PARTITION_MASK_MOD = 0  # int32(key & 0x7fffffff) % N   (CUDA-build branch of default_partition_fn)
PARTITION_FLOOR_MOD = 1  # key % N                       (CPU-build branch)
PARTITION_HASH = 2  # fmix64(key) % N               (opt-in, Zipf-balanced)


def partition(keys, num_shards, mode=PARTITION_MASK_MOD, n_dev=None):
  """default_partition_fn + dynamic_partition in one pass (PY/dynamic_embedding_variable.py:131-197).
  Returns owner-major keys, perm (original index of each output element) and device counts[num_shards].
  n_dev: optional device int64 scalar — only the first min(n, n_dev) keys are partitioned (the buffers
  keep length n; entries past sum(counts) are unspecified)."""
  flat = keys.contiguous().reshape(-1)
  n = flat.numel()
  dev = flat.device
  keys_out = torch.empty(n, dtype=torch.int64, device=dev)
  perm = torch.empty(n, dtype=torch.int32, device=dev)
  counts = torch.zeros(num_shards, dtype=torch.int64, device=dev)
  _capi.call("tfra_partition", _workspace(dev), n, None if n_dev is None else _ptr(n_dev), _ptr(flat), num_shards, mode,
             _ptr(keys_out), _ptr(perm), _ptr(counts), _stream(dev))
  return keys_out, perm, counts


def partition_by_owner(owner, num_shards):
  """dynamic_partition of range(n) by a caller-computed owner tensor (custom partitioners)."""
  owner = owner.reshape(-1).to(torch.int32).contiguous()
  n = owner.numel()
  dev = owner.device
  perm = torch.empty(n, dtype=torch.int32, device=dev)
  counts = torch.zeros(num_shards, dtype=torch.int64, device=dev)
  _capi.call("tfra_partition_by_owner", _workspace(dev), n, _ptr(owner), num_shards, _ptr(perm), _ptr(counts),
             _stream(dev))
  return perm, counts


def dynamic_partition(data, partitions, num_partitions):
  """tf.dynamic_partition on the device (K/dynamic_partition_op_gpu.cu.cc): `partitions` indexes the leading
  dims of `data`; returns `num_partitions` tensors, rows in input order.  Like the reference's GPU kernel,
  partition ids outside [0, num_partitions) are discarded (the CPU kernel raises).  The output shapes are
  data dependent: one host read of the counts, as for the TF op."""
  partitions = torch.as_tensor(partitions, device=data.device)
  if tuple(data.shape[:partitions.dim()]) != tuple(partitions.shape):
    raise ValueError("data.shape must start with partitions.shape: %s vs %s" % (list(data.shape), list(partitions.shape)))
  tail = tuple(data.shape[partitions.dim():])
  n = partitions.numel()
  width = 1
  for t in tail:
    width *= t
  if n == 0 or width == 0:
    counts = [0] * num_partitions if n == 0 else torch.bincount(
        partitions.reshape(-1).clamp(0, num_partitions).to(torch.int64), minlength=num_partitions + 1)[:num_partitions].tolist()
    return [torch.empty((c,) + tail, dtype=data.dtype, device=data.device) for c in counts]
  perm, counts = partition_by_owner(partitions.reshape(-1), num_partitions)
  counts = counts.tolist()
  rows = gather_rows(data.reshape(n, width), perm[:sum(counts)])
  return [r.reshape((-1,) + tail) for r in torch.split(rows, counts)]


def dynamic_stitch(indices, data):
  """tf.dynamic_stitch on the device (K/dynamic_stitch_op_gpu.cu.cc): merged[indices[m][i, ...]] =
  data[m][i, ...], first dim = max(index) + 1 (one host read).  Indices are expected to be distinct (what
  dynamic_partition of range(n) produces); rows no index names stay zero."""
  indices = [torch.as_tensor(i) for i in indices]
  data = [torch.as_tensor(d) for d in data]
  dev = data[0].device
  tail = tuple(data[0].shape[indices[0].dim():])
  for i, d in zip(indices, data):
    if tuple(d.shape[:i.dim()]) != tuple(i.shape) or tuple(d.shape[i.dim():]) != tail:
      raise ValueError("data[m].shape must be indices[m].shape + a common suffix")
  width = 1
  for t in tail:
    width *= t
  flat_idx = torch.cat([i.reshape(-1).to(device=dev, dtype=torch.int32) for i in indices]) if indices else None
  n_in = 0 if flat_idx is None else flat_idx.numel()
  n_out = int(flat_idx.max().item()) + 1 if n_in else 0
  out = torch.zeros((n_out,) + tail, dtype=data[0].dtype, device=dev)
  if n_in == 0 or width == 0:
    return out
  rows = torch.cat([d.reshape(-1, width) for d in data]).contiguous()
  _capi.call("tfra_scatter_rows", n_in, width * rows.element_size(), _ptr(rows), _ptr(flat_idx.contiguous()), _ptr(out),
             _stream(dev))
  return out


def select_lowest(keys, status, k):
  """The k keys with the lowest status (int32/int64), ties in input order (restrict policies)."""
  keys = keys.reshape(-1).to(torch.int64).contiguous()
  status = status.reshape(-1).contiguous()
  if status.dtype not in (torch.int32, torch.int64):
    raise TypeError("status must be int32 or int64")
  n = keys.numel()
  if status.numel() != n or k > n:
    raise ValueError("select_lowest: need one status per key and k <= n")
  out = torch.empty(k, dtype=torch.int64, device=keys.device)
  _capi.call("tfra_select_lowest", _workspace(keys.device), n, _ptr(keys), _ptr(status),
             4 if status.dtype == torch.int32 else 5, k, _ptr(out), _stream(keys.device))
  return out


COMBINERS = {"sum": 0, "mean": 1, "sqrtn": 2}


def sparse_segment_combine(rows, idx, seg, weights, combiner, n_rows):
  """out[r] = combine over {i: seg[i]==r} of weights[i]*rows[idx[i]]  (seg ascending; SparseSegment*)."""
  rows = rows.to(torch.float32).contiguous()
  idx = idx.to(torch.int32).contiguous()
  seg = seg.to(torch.int64).contiguous()
  if idx.numel() != seg.numel():       # T/math_ops_test.py:60-69
    raise ValueError("indices and segment_ids must have the same number of elements: %d vs %d" % (idx.numel(), seg.numel()))
  if weights is not None and torch.as_tensor(weights).numel() != idx.numel():
    raise ValueError("weights must have one element per index")
  w = None if weights is None else weights.to(torch.float32).contiguous()
  dim = rows.shape[-1]
  out = torch.empty((n_rows, dim), dtype=torch.float32, device=rows.device)
  _capi.call("tfra_sparse_segment_combine", _workspace(rows.device), idx.numel(), dim, _ptr(rows), _ptr(idx), _ptr(seg),
             _ptr(w), COMBINERS[combiner], n_rows, _ptr(out), _stream(rows.device))
  return out
