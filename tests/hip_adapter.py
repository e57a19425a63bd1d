"""Adapters that give the HIP table (through tfra_amd -> C ABI) the same numpy-facing op surface
as oracle.CpuTable, so KATs and differential tests run unchanged on both."""
import numpy as np
import torch

import tfra_amd.dynamic_embedding as de

_NP2T = {
    np.dtype(np.float32): torch.float32,
    np.dtype(np.float16): torch.float16,
    np.dtype(np.int8): torch.int8,
    np.dtype(np.int32): torch.int32,
    np.dtype(np.int64): torch.int64,
    np.dtype(np.float64): torch.float64,
}


def dev():
  return torch.device("cuda:0")


class HipTable:
  """numpy in / numpy out wrapper over de.CuckooHashTable (or de.HkvHashTable)."""

  def __init__(self, dim, dtype=np.float32, bf16=False, table_cls=None, **kw):
    self.dim = dim
    self.bf16 = bf16
    self.np_dtype = np.dtype(np.uint16) if bf16 else np.dtype(dtype)
    self.t_dtype = torch.bfloat16 if bf16 else _NP2T[np.dtype(dtype)]
    cls = table_cls or de.CuckooHashTable
    self.t = cls(torch.int64, self.t_dtype, default_value=torch.zeros(dim, dtype=self.t_dtype), device="cuda:0", dim=dim,
                 **kw)

  def _to_t(self, a, shape_like=None):
    a = np.asarray(a)
    if self.bf16:
      return torch.from_numpy(np.ascontiguousarray(a.astype(np.uint16)).view(np.int16)).to(dev()).view(torch.bfloat16)
    return torch.from_numpy(np.ascontiguousarray(a.astype(self.np_dtype))).to(dev())

  def _to_np(self, t):
    if self.bf16:
      return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
    return t.cpu().numpy()

  def _k(self, keys):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(keys, dtype=np.int64))).to(dev())

  def find(self, keys, defaults, return_exists=False):
    k = self._k(keys)
    d = self._to_t(defaults)
    r = self.t.lookup(k, dynamic_default_values=d, return_exists=return_exists)
    if return_exists:
      return self._to_np(r[0]).reshape(-1, self.dim), r[1].cpu().numpy().reshape(-1)
    return self._to_np(r).reshape(-1, self.dim)

  def insert(self, keys, values, clear=False):
    if clear:
      self.t.clear()
    self.t.insert(self._k(keys), self._to_t(values))

  def import_values(self, keys, values):
    self.insert(keys, values, clear=True)

  def accum(self, keys, vod, exists):
    self.t.accum(self._k(keys), self._to_t(vod), torch.from_numpy(np.asarray(exists, dtype=bool)).to(dev()))

  def remove(self, keys):
    self.t.remove(self._k(keys))

  def clear(self):
    self.t.clear()

  def size(self):
    return int(self.t.size().item())

  def export(self):
    k, v = self.t.export()
    return k.cpu().numpy(), self._to_np(v)

  def export_sorted(self):
    k, v = self.export()
    o = np.argsort(k, kind="stable")
    return k[o], v[o]


def hip_factory(**kw):
  def make(dim, dtype=np.float32, **kw2):
    return HipTable(dim, dtype, **{**kw, **kw2})
  return make
