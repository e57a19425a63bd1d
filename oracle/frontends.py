"""TEST INFRASTRUCTURE ONLY — NumPy restatement of the Python front-ends that bracket the
table ops in the reference (partition/stitch, unique/gather, sparse combiners, alltoall
routing).  PY = /root/reference/tensorflow_recommenders_addons/dynamic_embedding/python/ops
"""
import numpy as np


def default_partition_fn(keys, shard_num, gpu_mode=True):
  """PY/dynamic_embedding_variable.py:165-197.  int64 keys on CUDA builds:
  ``int32(key & 0x7fffffff) % N``; otherwise ``key % N`` (floor mod, like tf.math.mod)."""
  keys = np.asarray(keys, dtype=np.int64)
  if shard_num <= 1:
    return np.zeros(keys.shape, dtype=np.int32)
  if gpu_mode:
    return ((keys & 0x7FFFFFFF).astype(np.int32) % np.int32(shard_num)).astype(np.int32)
  return np.mod(keys, shard_num).astype(np.int32)


def dynamic_partition(data, partitions, num):
  """tf.dynamic_partition: `partitions` indexes the leading dims of `data`; stable split.  Partition ids
  outside [0, num) are DISCARDED — the behaviour of the reference's GPU kernel
  (K/dynamic_partition_op_gpu.cu.cc, T/dynamic_partition_op_test.py:221-283); its CPU kernel raises."""
  data = np.asarray(data)
  partitions = np.asarray(partitions)
  tail = data.shape[partitions.ndim:]
  flat = data.reshape((partitions.size,) + tail)
  p = partitions.reshape(-1)
  return [flat[p == q] for q in range(num)]


def make_partition(data, partition_index, shard_num):
  """PY/dynamic_embedding_variable.py:131-154 -> (partitions, indices)."""
  if shard_num <= 1:
    return [np.asarray(data)], None
  parts = dynamic_partition(data, partition_index, shard_num)
  idx = dynamic_partition(np.arange(len(data)), partition_index, shard_num)
  return parts, idx


def dynamic_stitch(indices, values):
  """tf.dynamic_stitch: merged[indices[m][i, ...]] = values[m][i, ...]; first dim = max(index) + 1
  (PY/dynamic_embedding_variable.py:157-162, T/dynamic_stitch_op_test.py)."""
  indices = [np.asarray(i) for i in indices]
  values = [np.asarray(v) for v in values]
  tail = values[0].shape[indices[0].ndim:]
  n = max([int(i.max()) + 1 for i in indices if i.size] + [0])
  out = np.zeros((n,) + tail, dtype=values[0].dtype)
  for i, v in zip(indices, values):
    out[i.reshape(-1)] = v.reshape((-1,) + tail)
  return out


def unique(ids):
  """tf.unique: y in order of first occurrence, idx such that y[idx] == ids."""
  ids = np.asarray(ids).reshape(-1)
  _, first, inv = np.unique(ids, return_index=True, return_inverse=True)
  order = np.argsort(first, kind="stable")
  rank = np.empty_like(order)
  rank[order] = np.arange(order.size)
  return ids[np.sort(first)], rank[inv].astype(np.int32)


def embedding_lookup_unique(table, ids, defaults):
  """PY/dynamic_embedding_ops.py:64-117: unique -> lookup -> gather -> reshape."""
  ids = np.asarray(ids)
  u, idx = unique(ids)
  rows = table.find(u, defaults)
  return rows[idx].reshape(ids.shape + (rows.shape[1],))


def embedding_lookup_sparse(table, sp_indices_row, sp_values, defaults, combiner="mean",
                            sp_weights=None, num_rows=None):
  """PY/dynamic_embedding_ops.py:120-293: unique ids -> lookup -> (weights) ->
  segment sum / mean / sqrtn over the sparse row ids (sorted segment ids)."""
  seg = np.asarray(sp_indices_row, dtype=np.int64)
  ids = np.asarray(sp_values, dtype=np.int64)
  u, idx = unique(ids)
  emb = table.find(u, defaults).astype(np.float32)[idx]
  w = np.ones(len(ids), np.float32) if sp_weights is None else np.asarray(sp_weights, np.float32)
  n = int(seg.max()) + 1 if num_rows is None else num_rows
  out = np.zeros((n, emb.shape[1]), np.float32)
  np.add.at(out, seg, emb * w[:, None])
  if combiner == "sum":
    return out
  wsum = np.zeros(n, np.float32)
  if combiner == "mean":
    np.add.at(wsum, seg, w)
  elif combiner == "sqrtn":
    np.add.at(wsum, seg, w * w)
    wsum = np.sqrt(wsum)
  else:
    raise ValueError("combiner must be one of 'mean', 'sqrtn' or 'sum'")
  with np.errstate(invalid="ignore", divide="ignore"):
    res = out / wsum[:, None]
  res[wsum == 0] = 0  # empty segments stay zero rows
  return res


def safe_embedding_lookup_sparse(table, indices, ids, dense_shape, defaults, weights=None, combiner="mean",
                                 default_id=None):
  """PY/dynamic_embedding_ops.py:296-430 (TFRA's variant: ids are never pruned): flatten the leading dims,
  drop entries with weight <= 0 unless combiner == "sum", combine per row, empty rows -> zeros or the
  embedding of `default_id`, restore the leading dims."""
  indices = np.asarray(indices, dtype=np.int64).reshape(len(ids), -1)
  ids = np.asarray(ids, dtype=np.int64)
  lead = [int(x) for x in dense_shape[:-1]]
  rows = np.ravel_multi_index(tuple(indices[:, d] for d in range(len(lead))), lead) if len(ids) else np.zeros(0, np.int64)
  n = int(np.prod(lead))
  w = None if weights is None else np.asarray(weights, np.float32)
  if w is not None and combiner != "sum":
    keep = w > 0
    rows, ids, w = rows[keep], ids[keep], w[keep]
  dim = np.asarray(defaults).reshape(-1).size
  if len(ids):
    res = embedding_lookup_sparse(table, rows, ids, defaults, combiner=combiner, sp_weights=w, num_rows=n)
  else:
    res = np.zeros((n, dim), np.float32)
  if default_id is not None:
    empty = np.ones(n, bool)
    empty[rows] = False
    res[empty] = table.find(np.array([default_id], np.int64), defaults).astype(np.float32)[0]
  return res.reshape(tuple(lead) + (res.shape[-1],))


def sharded_lookup(tables, ids, defaults_fn, partition_fn=default_partition_fn):
  """Variable.lookup over N shards: partition -> per-shard find -> stitch
  (PY/dynamic_embedding_variable.py:933-986)."""
  ids = np.asarray(ids, dtype=np.int64).reshape(-1)
  n = len(tables)
  part = partition_fn(ids, n)
  kp, ki = make_partition(ids, part, n)
  vals = [tables[i].find(kp[i], defaults_fn(len(kp[i]))) for i in range(n)]
  if n == 1:
    return vals[0]
  return dynamic_stitch(ki, vals)


def alltoall_lookup_model(rank_tables, ids_per_rank, defaults, partition_fn=default_partition_fn):
  """Single-process model of HvdVariable.__alltoall_embedding_lookup__
  (PY/shadow_embedding_ops.py:397-447): every rank partitions its ids by owner, owners look
  the ids up in their local shard, rows return to the asking rank in input order."""
  world = len(rank_tables)
  out = []
  for r in range(world):
    ids = np.asarray(ids_per_rank[r], dtype=np.int64).reshape(-1)
    part = partition_fn(ids, world)
    rows = np.zeros((ids.size, rank_tables[0].dim), dtype=rank_tables[0].dtype)
    for owner in range(world):
      sel = np.nonzero(part == owner)[0]
      if sel.size:
        rows[sel] = rank_tables[owner].find(ids[sel], defaults)
    out.append(rows)
  return out


# ---- size restriction (PY/restrict_policies.py) ---------------------------------------------------
def restrict_select(keys, status, num_reserved):
  """Keys a shard removes: `top_k(-status, n - reserved)` then gather (PY/restrict_policies.py:213-223,
  341-351).  top_k(sorted=False) leaves the choice among EQUAL statuses unspecified; this restatement
  (and the HIP path) resolves it by export order (stable)."""
  keys = np.asarray(keys, dtype=np.int64).reshape(-1)
  status = np.asarray(status).reshape(-1)
  k = max(keys.size - int(num_reserved), 0)
  order = np.argsort(status.astype(np.int64), kind="stable")
  return keys[order[:k]]


class RestrictPolicyOracle:
  """Timestamp / Frequency policies over plain dicts (PY/restrict_policies.py:118-361): `apply_update`
  and `apply_restriction(num_reserved, trigger)` for ONE shard; `table` is any dict-like key -> row."""

  def __init__(self, kind):
    assert kind in ("timestamp", "frequency")
    self.kind = kind
    self.status = {}

  def apply_update(self, ids, now=None):
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    if self.kind == "timestamp":
      for k in ids:
        self.status[int(k)] = int(now)                # :168-177 every id gets the same fresh stamp
    else:
      old = {int(k): self.status.get(int(k), 0) for k in ids}   # :292-297 lookup (default 0) for the batch
      for k in ids:
        self.status[int(k)] = old[int(k)] + 1         # +1 once per call, duplicates included

  def apply_restriction(self, table, num_reserved, trigger=None):
    trigger = num_reserved if trigger is None else trigger
    if len(table) <= trigger:                        # :202 cond(size > trigger)
      return []
    ks = np.fromiter(self.status.keys(), dtype=np.int64, count=len(self.status))
    st = np.fromiter(self.status.values(), dtype=np.int64, count=len(self.status))
    gone = restrict_select(ks, st, num_reserved)
    for k in gone:
      table.pop(int(k), None)
      self.status.pop(int(k), None)
    return [int(k) for k in gone]


# ---- the metric's step on a sharded table (csrc/tfra_aroute.hip restated) ----------------------------
def route_plan(ids, world, partition_fn=default_partition_fn):
  """What the route plan of one batch holds (tfra_aroute.hip: routeplan_insert_kernel + routeplan_emit_kernel): the distinct
  ids grouped by owner (order inside a group unspecified: here sorted), the LAST position of each, the position -> row map and
  the per-owner counts.  PY/shadow_embedding_ops.py:316,397-422 does unique then dynamic_partition."""
  ids = np.asarray(ids, dtype=np.int64).reshape(-1)
  uniq, inv = np.unique(ids, return_inverse=True)
  last = np.zeros(uniq.size, dtype=np.int64)
  np.maximum.at(last, inv, np.arange(ids.size))
  owner = partition_fn(uniq, world)
  order = np.argsort(owner, kind="stable")
  row_of_uniq = np.empty(uniq.size, dtype=np.int64)
  row_of_uniq[order] = np.arange(uniq.size)
  counts = np.bincount(owner, minlength=world).astype(np.int64)
  return uniq[order], last[order], row_of_uniq[inv], counts


def routed_assign_model(table, ids_per_rank, values_per_rank, defaults):
  """ONE table, one step of every rank: all lookups first, then rank 0's insert_or_assign, rank 1's, ... (the last occurrence of
  a repeated key wins: sequential insert).  Returns the rows every rank's lookup returns.  What RoutedAssignStep must equal
  (PY/shadow_embedding_ops.py:397-447 for the lookup, PY/dynamic_embedding_variable.py:772-800 for the sharded upsert)."""
  rows = [table.find(np.asarray(i, np.int64).reshape(-1), defaults) for i in ids_per_rank]
  for i, v in zip(ids_per_rank, values_per_rank):
    if v is not None:
      table.insert(np.asarray(i, np.int64).reshape(-1), v)
  return rows
