"""The TFRA-side Python of the fused MI355X ops (tf_ops/fused_ops_rocm.cc): what a maintainer adds next to
`tensorflow_recommenders_addons/dynamic_embedding/python/ops/` so that an existing TFRA training graph takes the fused paths.

TensorFlow is absent from this repository's image, so this module is SOURCE for the integration (syntax-checked by
tests/test_tf_shim.py; it imports TensorFlow and TFRA lazily, inside the functions).  Each function names the reference code it
swaps out (R = tensorflow_recommenders_addons/dynamic_embedding/python/ops):

  load_ops()                          tf.load_op_library of the _hkv_ops.so built from hkv_ops_rocm.cc + fused_ops_rocm.cc
                                      (R/../../utils/resource_loader.py:104-120 loads the reference's the same way)
  FusedHkvHashTable                   HkvHashTable (R/hkv_hashtable_ops.py) whose creator op carries `optimizer_slots`
  embedding_lookup_unique_fused       embedding_lookup_unique (R/dynamic_embedding_ops.py:99-117)
  patch_optimizer_apply(optimizer)    the `_apply_op` of DynamicEmbeddingOptimizer (R/dynamic_embedding_optimizer.py:165-204) and
                                      create_slots (:870-958)
  LookupAssignLoop                    a streaming Find / Insert loop on one table as ONE launch per step
  AllToAllRoute                       HvdAllToAllEmbedding's lookup-time exchange (R/shadow_embedding_ops.py:397-447)
"""

_OPS = None


def load_ops(path=None):
  """The generated wrappers of every op of _hkv_ops.so (`tfra_hkv_hash_table_find`, ..., and the fused ones below)."""
  global _OPS
  if _OPS is None:
    import tensorflow as tf
    from tensorflow_recommenders_addons.utils.resource_loader import get_path_to_datafile
    _OPS = tf.load_op_library(path or get_path_to_datafile("dynamic_embedding/core/_hkv_ops.so"))
  return _OPS


# ------------------------------------------------------------------------------------------------ the table with slots
def make_fused_table_class():
  """-> FusedHkvHashTable: R/hkv_hashtable_ops.py's HkvHashTable with ONE change — `_create_resource` issues
  TFRA>HkvHashTableOfTensorsWithSlots, so the rows are [param | slot_1 .. slot_S] and the fused optimizer ops find the slots
  next to the parameter (create_slots then returns views, see patch_optimizer_apply)."""
  from tensorflow_recommenders_addons.dynamic_embedding.python.ops.hkv_hashtable_ops import HkvHashTable

  class FusedHkvHashTable(HkvHashTable):

    def __init__(self, *args, optimizer_slots=0, slot_init=(), **kwargs):
      self._optimizer_slots = int(optimizer_slots)
      self._slot_init = [float(x) for x in slot_init]
      super().__init__(*args, **kwargs)

    def _create_resource(self):
      ops = load_ops()
      # the reference's call (R/hkv_hashtable_ops.py `_create_resource`) + the two attrs
      table_ref = ops.tfra_hkv_hash_table_of_tensors_with_slots(
          shared_name=self._shared_name, use_node_name_sharing=self._checkpoint and self._shared_name is None,
          key_dtype=self._key_dtype, value_dtype=self._value_dtype, value_shape=self._default_value.get_shape(),
          init_capacity=self._init_capacity, max_capacity=self._max_capacity, max_hbm_for_vectors=self._max_hbm_for_vectors,
          step_per_epoch=self._step_per_epoch, strategy=self._strategy, reserved_key_start_bit=self._reserved_key_start_bit,
          optimizer_slots=self._optimizer_slots, slot_init=self._slot_init, name=self._name)
      import tensorflow as tf
      self._table_name = table_ref.op.name.split("/")[-1] if not tf.executing_eagerly() else None
      return table_ref

  return FusedHkvHashTable


# ------------------------------------------------------------------------------------------------ lookup
def embedding_lookup_unique_fused(params, ids, name=None):
  """embedding_lookup_unique (R/dynamic_embedding_ops.py:99-117) for a single-shard FusedHkvHashTable variable:

      unique_ids, idx = array_ops.unique(ids_flat)            # :99    host reads the count: it is an output SHAPE
      unique_embeddings = de.embedding_lookup(params, unique_ids, ...)   # :100   Find op
      embeddings_flat = array_ops.gather(unique_embeddings, idx)         # :109

  becomes ONE op whose outputs all have the upper-bound length B = ids.size (no host read; `num_unique` stays on the device):
  tfra_table_find_unique — the lookup of all B ids and their de-duplication in one kernel launch.  Returns (embeddings [ids.shape + dim], unique_ids [B], idx [B],
  num_unique): the last three are what TrainableWrapper keeps for the backward pass (the optimizer op below takes the ids WITH
  repeats, so a graph that only trains needs `embeddings` alone)."""
  import tensorflow as tf
  ops = load_ops()
  table = params.tables[0]
  with tf.name_scope(name or "EmbeddingLookupUniqueFused"):
    ids = tf.convert_to_tensor(ids, dtype=tf.int64)
    default_row = tf.zeros([params.dim], dtype=params.value_dtype) if params.initializer is None else tf.reshape(
        params.initializer(shape=[params.dim], dtype=params.value_dtype), [params.dim])
    values, unique_ids, idx, num_unique = ops.tfra_hkv_hash_table_embedding_lookup(
        table.resource_handle, tf.reshape(ids, [-1]), default_row, key_dtype=tf.int64, value_dtype=params.value_dtype)
    values = tf.reshape(values, tf.concat([tf.shape(ids), [params.dim]], 0))
    values.set_shape(ids.get_shape().concatenate([params.dim]))
    return values, unique_ids, idx, num_unique


# ------------------------------------------------------------------------------------------------ optimizer
_SLOTS = {"SGD": 0, "GradientDescent": 0, "Adam": 2, "Adagrad": 1, "Ftrl": 2}


def patch_optimizer_apply(optimizer):
  """DynamicEmbeddingOptimizer(optimizer) (R/dynamic_embedding_optimizer.py) wraps `apply_gradients`; per sparse variable its
  `_apply_op` runs (:165-204):

      _slots = [self.get_slot(var, _s) for _s in self.get_slot_names()]        # :166  S more hash tables (create_slots :870-958)
      v0 = var.read_value(do_prefetch=...); s0 = [_s.read_value() for _s in _slots]      # :172-176  (1+S) Find ops
      _apply_op = self._resource_apply_sparse_duplicate_indices(grad.values, var, grad.indices, **apply_kwargs)   # :185 unique + segment_sum + ResourceApply*
      _after = group([var.update_op(v0=v0)] + [_s.update_op(v0=s0[si]) ...])                                      # :188-190 (1+S) Insert ops

  With a FusedHkvHashTable (rows [p | slots]) the whole block is ONE op per variable: TFRA>HkvHashTableApplySparse{Sgd,Adam,Adagrad,Ftrl}
  on the ids WITH their repeats (`var.ids` of the TrainableWrapper, R/embedding_weights.py) and the gradient rows.  Optimizers without
  a fused rule keep the reference sequence."""
  import tensorflow as tf
  ops = load_ops()
  kind = type(optimizer).__name__
  if kind not in _SLOTS:
    return optimizer   # Generic path: the reference's sequence through the 14 table ops

  def _fused_apply_op(grad, var, apply_state=None):
    # `var` = the TrainableWrapper of one lookup; var.params.tables[0] = the FusedHkvHashTable shard; var.ids = the ids looked up
    table = var.params.tables[0]
    ids = tf.reshape(var.ids, [-1])
    g = tf.reshape(grad.values if isinstance(grad, tf.IndexedSlices) else grad, [-1, var.params.dim])
    if isinstance(grad, tf.IndexedSlices):      # gradient rows of the UNIQUE lookups: back to the ids they belong to
      ids = tf.gather(ids, grad.indices)
    default_row = tf.zeros([var.params.dim], tf.float32)
    lr = tf.cast(optimizer._decayed_lr(tf.float32), tf.float32)
    h = table.resource_handle
    if kind in ("SGD", "GradientDescent"):
      return ops.tfra_hkv_hash_table_apply_sparse_sgd(h, ids, g, default_row, lr)
    if kind == "Adam":
      t = tf.cast(optimizer.iterations + 1, tf.float32)
      b1, b2 = tf.cast(optimizer._get_hyper("beta_1"), tf.float32), tf.cast(optimizer._get_hyper("beta_2"), tf.float32)
      return ops.tfra_hkv_hash_table_apply_sparse_adam(h, ids, g, default_row, lr, tf.pow(b1, t), tf.pow(b2, t), b1, b2,
                                                       tf.cast(optimizer.epsilon, tf.float32))
    if kind == "Adagrad":
      return ops.tfra_hkv_hash_table_apply_sparse_adagrad(h, ids, g, default_row, lr, tf.cast(optimizer.epsilon, tf.float32), use_epsilon=True)
    return ops.tfra_hkv_hash_table_apply_sparse_ftrl(h, ids, g, default_row, lr,
                                                     tf.cast(optimizer._get_hyper("l1_regularization_strength"), tf.float32),
                                                     tf.cast(optimizer._get_hyper("l2_regularization_strength"), tf.float32),
                                                     tf.cast(optimizer._get_hyper("learning_rate_power"), tf.float32))

  optimizer._tfra_fused_apply_op = _fused_apply_op          # called by the patched `_apply_op` instead of lines :165-190
  optimizer._tfra_fused_slots = _SLOTS[kind]                # FusedHkvHashTable(optimizer_slots=...) for every trainable table
  # slot_init as TensorFlow creates the slot variables: Adam m = v = 0; Adagrad accumulator = initial_accumulator_value;
  # FTRL accumulator = initial_accumulator_value, linear = 0
  iav = float(getattr(optimizer, "_initial_accumulator_value", 0.1))
  optimizer._tfra_fused_slot_init = {"Adam": (0.0, 0.0), "Adagrad": (iav,), "Ftrl": (iav, 0.0)}.get(kind, ())
  return optimizer


# ------------------------------------------------------------------------------------------------ Find + Insert as one launch
class LookupAssignLoop:
  """A loop that alternates Find and Insert on ONE bounded table (an embedding cache being refreshed, a parameter-server shard):

      rows_i = table.lookup(ids_i)            # TFRA>HkvHashTableFind
      table.insert(ids_i, new_rows_i)         # TFRA>HkvHashTableInsert            (K/hkv_hashtable_op_gpu.cu.cc:182-290)

  as TFRA>HkvHashTableLookupAssignStep: the write-back of batch i runs in the launch that looks batch i+1 up (results identical
  to the two ops in order).  ids of the next two batches come from the input pipeline's prefetch buffer."""

  def __init__(self, table, dim, value_dtype):
    self.ops, self.table, self.dim, self.vdt = load_ops(), table, dim, value_dtype
    self._first = True

  def step(self, ids, prev_values, ids_next, ids_next2, default_row):
    import tensorflow as tf
    empty = tf.zeros([0, self.dim], self.vdt)
    rows, exists = self.ops.tfra_hkv_hash_table_lookup_assign_step(
        self.table.resource_handle, ids, default_row, empty if self._first else prev_values, ids_next, ids_next2,
        value_dtype=self.vdt, ids_were_announced=not self._first)
    self._first = False
    return rows, exists

  def flush(self, prev_values):
    self._first = True
    return self.ops.tfra_hkv_hash_table_lookup_assign_flush(self.table.resource_handle, prev_values, value_dtype=self.vdt)


# ------------------------------------------------------------------------------------------------ the multi-GPU route
class AllToAllRoute:
  """HvdAllToAllEmbedding (R/../keras/layers/embedding.py:545-594 -> R/shadow_embedding_ops.py:397-447) for one table shard per
  rank: `__alltoall_embedding_lookup__`'s unique -> partition -> hvd.alltoall(ids) -> local lookup -> hvd.alltoall(rows) -> stitch,
  and the mirrored gradient exchange, issued by the C driver with the id-only half prepared up to three batches ahead."""

  def __init__(self, table, rank, world, max_batch, librccl_path="librccl.so"):
    import horovod.tensorflow as hvd
    import tensorflow as tf
    self.ops = load_ops()
    ids = self.ops.tfra_rccl_unique_id(librccl_path=librccl_path) if rank == 0 else tf.zeros([256], tf.uint8)
    ids = hvd.broadcast(ids, root_rank=0)     # 2 x 128 bytes through the host framework
    self.handle = self.ops.tfra_route_create(table.resource_handle, ids, rank=rank, world=world, partition_mode=0,
                                             max_batch=max_batch, librccl_path=librccl_path,
                                             shared_name="route_" + str(table.name))

  def feed(self, ids):                         # from the dataset's prefetch stage, three batches ahead of lookup()
    return self.ops.tfra_route_feed(self.handle, ids)

  def lookup(self, default_row):               # rows of the OLDEST fed batch, in its own order
    return self.ops.tfra_route_lookup(self.handle, default_row)

  def apply_adam(self, grads, default_row, lr, beta1_power, beta2_power, beta1, beta2, epsilon):
    return self.ops.tfra_route_apply_adam(self.handle, grads, default_row, lr, beta1_power, beta2_power, beta1, beta2, epsilon)

  def apply_sgd(self, grads, default_row, lr):
    return self.ops.tfra_route_apply_sgd(self.handle, grads, default_row, lr)

  def apply_adagrad(self, grads, default_row, lr, epsilon):
    return self.ops.tfra_route_apply_adagrad(self.handle, grads, default_row, lr, epsilon, use_epsilon=True)

  def apply_ftrl(self, grads, default_row, lr, l1, l2, lr_power):
    return self.ops.tfra_route_apply_ftrl(self.handle, grads, default_row, lr, l1, l2, lr_power)


class AllToAllAssignRoute:
  """The metric's step — lookup(ids) + insert_or_assign(ids, rows), repeats: the last occurrence wins — on one table shard per rank:
  the lookup half is `__alltoall_embedding_lookup__` (R/shadow_embedding_ops.py:397-447), the write-back half the sharded
  `Variable.upsert` (R/dynamic_embedding_variable.py:772-800: keys AND values partitioned by owner), issued by the C driver
  (tfra_assign_route_*): per batch two launches for everything that depends on the ids alone, ahead of the step; per step
  gather -> alltoall(values) -> ONE launch at the owner (lookup of this batch + write-back of the previous one) -> alltoall(rows)
  -> gather.

      route = AllToAllAssignRoute(table, hvd.rank(), hvd.size(), max_batch, value_dtype)
      for k in range(5): route.feed(next(batches))            # the dataset's prefetch stage: five batches ahead
      rows = route.step(default_row, zero_rows)               # first step: nothing to write back (prev_values is ignored)
      loop: route.feed(next(batches)); rows = route.step(default_row, new_rows_of_the_previous_batch)
      route.flush(new_rows_of_the_last_batch)"""

  def __init__(self, table, rank, world, max_batch, value_dtype, librccl_path="librccl.so"):
    import horovod.tensorflow as hvd
    import tensorflow as tf
    self.ops = load_ops()
    self.vdt = value_dtype
    ids = self.ops.tfra_rccl_unique_id(librccl_path=librccl_path) if rank == 0 else tf.zeros([256], tf.uint8)
    ids = hvd.broadcast(ids, root_rank=0)     # 2 x 128 bytes through the host framework
    self.handle = self.ops.tfra_assign_route_create(table.resource_handle, ids, rank=rank, world=world, partition_mode=0,
                                                    max_batch=max_batch, librccl_path=librccl_path,
                                                    shared_name="assign_route_" + str(table.name))

  def feed(self, ids):
    return self.ops.tfra_assign_route_feed(self.handle, ids)

  def step(self, default_row, prev_values):    # rows of the OLDEST fed batch; prev_values = what the previous step's batch writes back
    return self.ops.tfra_assign_route_step(self.handle, default_row, prev_values, value_dtype=self.vdt)

  def flush(self, prev_values):
    return self.ops.tfra_assign_route_flush(self.handle, prev_values, value_dtype=self.vdt)
