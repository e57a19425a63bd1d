"""Size restriction of a dynamic-embedding Variable.

Mirror of `PY/restrict_policies.py:36-361`: a policy owns a *status* Variable (same keys as the
target, one int32 per key: last-seen timestamp or occurrence count), `apply_update(ids)` refreshes
it for the ids of a training step (`PY/embedding_weights.py:441-442` calls it from the write-back),
and `apply_restriction(num_reserved, trigger=...)` removes all but the `num_reserved` most
recent / most frequent keys once `size > trigger`.

The reference does the selection in Python graph ops per shard: export -> top_k(-status, size -
reserved) -> gather -> remove from the target, the status table and every slot variable
(`:205-230,332-358`).  Here the selection is one device call (`tfra_select_lowest`: stable radix sort
of the exported status, k lowest keys) followed by `tfra_table_erase`; optimizer slots are state
fields of the target's own rows (`aux_fields`), so they go with the row.
"""
import time

import torch

from . import device_ops


class RestrictPolicy:
  """PY/restrict_policies.py:36-115"""

  def __init__(self, var):
    from .variable import Variable
    if not isinstance(var, Variable):
      raise TypeError("parameter var type should be dynamic_embedding.Variable.")
    self.var = var
    self.params_in_slots = []
    self._restrict_var = None

  def apply_update(self, ids):
    raise NotImplementedError

  def apply_restriction(self, num_reserved, **kwargs):
    raise NotImplementedError

  @property
  def status(self):
    raise NotImplementedError

  def _track_params_from_optimizer_slots(self, slots):
    """PY/restrict_policies.py:102-115.  Slot variables that are separate tables (not views of the
    target's rows) are restricted together with the target."""
    for s in slots:
      p = getattr(s, "params", s)
      if p is not self.var and all(p is not q for q in self.params_in_slots) and hasattr(p, "tables"):
        self.params_in_slots.append(p)

  # -- shared machinery ------------------------------------------------------------------------
  def _make_status(self, suffix):
    from .variable import Variable
    return Variable(key_dtype=self.var.key_dtype, value_dtype=torch.int32, dim=1, devices=self.var.devices,
                    partitioner=self.var.partition_fn, name=self.var.name + suffix, trainable=False,
                    init_size=self.var.init_size * self.var.shard_num, kv_creator=self.var.kv_creator)

  def _check_args(self, num_reserved, kwargs):
    if not isinstance(num_reserved, int):
      raise TypeError("num_reserved should be integer.")
    if num_reserved < 0:
      raise ValueError("num_reserved should be non-negative.")
    trigger = kwargs.get("trigger", num_reserved)
    if not isinstance(trigger, int):
      raise TypeError("trigger should be integer.")
    return trigger

  def _restrict(self, num_reserved, trigger):
    """`cond(size > trigger, _cond_restrict_fn, no_op)` (PY/restrict_policies.py:202-230)."""
    if int(self.var.size()) <= trigger:
      return 0
    status = self._restrict_var
    partial_reserved = int(num_reserved / status.shard_num)
    removed = 0
    for i, st in enumerate(status.tables):
      keys, vals = st.export()
      n = keys.numel()
      k = max(n - partial_reserved, 0)
      if k == 0:
        continue
      victims = device_ops.select_lowest(keys, vals.reshape(-1), k)
      self.var.tables[i].remove(victims.to(self.var.tables[i]._device))
      st.remove(victims)
      for sp in self.params_in_slots:
        sp.tables[i].remove(victims.to(sp.tables[i]._device))
      removed += k
    return removed


class TimestampRestrictPolicy(RestrictPolicy):
  """Oldest-out-first (PY/restrict_policies.py:118-233)."""

  def __init__(self, var):
    super().__init__(var)
    self.tstp_var = self._make_status("/timestamp")
    self._restrict_var = self.tstp_var

  def apply_update(self, ids):
    """Every id of the step gets the current wall-clock second (`:159-179`)."""
    keys = torch.as_tensor(ids, device=self.var._primary).reshape(-1)
    now = int(time.time()) & 0x7fffffff
    fresh = torch.full((keys.numel(), 1), now, dtype=torch.int32, device=keys.device)
    self.tstp_var.upsert(keys, fresh)

  def apply_restriction(self, num_reserved, **kwargs):
    trigger = self._check_args(num_reserved, kwargs)
    return self._restrict(num_reserved, trigger)

  @property
  def status(self):
    return self.tstp_var


class FrequencyRestrictPolicy(RestrictPolicy):
  """Least-frequent-out-first (PY/restrict_policies.py:236-361)."""

  def __init__(self, var):
    super().__init__(var)
    self.init_count = 0
    self.freq_var = self._make_status("/frequency")
    self._restrict_var = self.freq_var

  def apply_update(self, ids):
    """count[id] = lookup(id, default 0) + 1; an id repeated inside one call still counts once,
    exactly as the reference's lookup / +1 / insert sequence does (`:278-303`)."""
    keys = torch.as_tensor(ids, device=self.var._primary).reshape(-1)
    counts = self.freq_var.lookup(keys)
    self.freq_var.upsert(keys, counts + 1)

  def apply_restriction(self, num_reserved, **kwargs):
    trigger = self._check_args(num_reserved, kwargs)
    return self._restrict(num_reserved, trigger)

  @property
  def status(self):
    return self.freq_var
