"""TEST INFRASTRUCTURE ONLY.

CPU checkers for the dynamic-embedding hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``recommenders-addons_amd/``) never does.

Two engines behind one ctypes binding (same C signatures):

* ``kind="port"``       ``oracle/libtfra_oracle.so`` — plain-C restatement
  (``oracle/tfra_oracle.c``), always buildable.
* ``kind="reference"``  ``oracle/_ref/libtfra_ref.so`` — the reference's own vendored
  ``cuckoohash_map.hh`` compiled in place from ``/root/reference`` behind a raw-pointer
  restatement of ``TableWrapperOptimized`` (``oracle/ref_shim.cc``).  float32 / int8 /
  int32 / int64 / float64 values only (half/bfloat16 need Eigen, absent here).

Parity status: PINNED — the port is checked against the reference's inline KATs
 and against the real reference engine on seeded op sequences
(``tests/golden/*.npz``, generator ``tests/golden/make_golden.py``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# dtype codes shared with include/tfra_mi355x.h
DT_F32, DT_F16, DT_BF16, DT_I8, DT_I32, DT_I64, DT_F64 = range(7)
_NP2DT = {
    np.dtype(np.float32): DT_F32,
    np.dtype(np.float16): DT_F16,
    np.dtype(np.int8): DT_I8,
    np.dtype(np.int32): DT_I32,
    np.dtype(np.int64): DT_I64,
    np.dtype(np.float64): DT_F64,
}


def dtype_code(dtype, bf16=False):
  if bf16:
    return DT_BF16
  return _NP2DT[np.dtype(dtype)]


def build(force=False):
  """Compile both checkers (``make -C oracle``). The reference leg is skipped, keeping any
  prebuilt ``_ref/libtfra_ref.so``, where ``/root/reference`` does not exist."""
  if force:
    subprocess.check_call(["make", "-C", _HERE, "clean"])
  subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def _lib_path(kind):
  if kind == "port":
    return os.path.join(_HERE, "libtfra_oracle.so")
  if kind == "reference":
    return os.path.join(_HERE, "_ref", "libtfra_ref.so")
  raise ValueError(kind)


def available(kind):
  return os.path.exists(_lib_path(kind))


_LIBS = {}


def _load(kind):
  if kind in _LIBS:
    return _LIBS[kind]
  path = _lib_path(kind)
  if not os.path.exists(path):
    build()
  lib = ctypes.CDLL(path)
  p = "tfra_oracle_" if kind == "port" else "tfra_ref_"
  c = ctypes
  sig = {
      "create": (c.c_void_p, [c.c_int, c.c_longlong, c.c_ulonglong]),
      "destroy": (None, [c.c_void_p]),
      "find": (None, [c.c_void_p, c.c_longlong, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int,
                      c.c_void_p, c.c_int]),
      "insert": (None, [c.c_void_p, c.c_longlong, c.c_void_p, c.c_void_p, c.c_int, c.c_int]),
      "accum": (None, [c.c_void_p, c.c_longlong, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int]),
      "remove": (None, [c.c_void_p, c.c_longlong, c.c_void_p]),
      "clear": (None, [c.c_void_p]),
      "size": (c.c_ulonglong, [c.c_void_p]),
      "dump": (c.c_ulonglong, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_ulonglong,
                               c.c_ulonglong]),
  }
  fns = {}
  for name, (res, args) in sig.items():
    f = getattr(lib, p + name)
    f.restype = res
    f.argtypes = args
    fns[name] = f
  _LIBS[kind] = fns
  return fns


def _ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class CpuTable:
  """One CPU dynamic-embedding table with the reference op semantics
  (``CuckooHashTableOfTensors``, R/kernels/cuckoo_hashtable_op.cc:184-308).

  Values of dtype bfloat16 are carried as ``np.uint16`` bit patterns with ``bf16=True``.
  """

  def __init__(self, dim, dtype=np.float32, kind="port", init_size=0, threads=1, bf16=False):
    self._f = _load(kind)
    self.kind = kind
    self.dim = int(dim)
    self.bf16 = bf16
    self.dtype = np.dtype(np.uint16) if bf16 else np.dtype(dtype)
    self.threads = threads
    self._code = dtype_code(self.dtype, bf16)
    self._h = self._f["create"](self._code, self.dim, int(init_size))
    if not self._h:
      raise ValueError("dtype %s not supported by the %s engine" % (self.dtype, kind))

  def __del__(self):
    if getattr(self, "_h", None):
      self._f["destroy"](self._h)
      self._h = None

  def _keys(self, keys):
    return np.ascontiguousarray(np.asarray(keys, dtype=np.int64).reshape(-1))

  def _vals(self, values, keys):
    v = np.ascontiguousarray(np.asarray(values, dtype=self.dtype))
    want = tuple(np.shape(keys)) + (self.dim,)
    if v.shape != want:
      # CheckKeyAndValueTensorsForInsert: values.shape == keys.shape + value_shape
      # (R/kernels/cuckoo_hashtable_op.cc:640-660; K6)
      raise ValueError("Expected shape %s for values, got %s" % (list(want), list(v.shape)))
    return v.reshape(-1, self.dim)

  def find(self, keys, defaults, return_exists=False):
    k = self._keys(keys)
    n = k.size
    d = np.ascontiguousarray(np.asarray(defaults, dtype=self.dtype))
    out = np.empty((n, self.dim), dtype=self.dtype)
    # is_full_default = (value_flat.size() == default_flat.size())  cuckoo_hashtable_op.cc:48-50
    full = int(out.size == d.size)
    if not full and d.size < self.dim:
      raise ValueError("default value needs at least dim elements")
    ex = np.zeros(n, dtype=np.uint8) if return_exists else None
    self._f["find"](self._h, n, _ptr(k), _ptr(out), _ptr(d), full, _ptr(ex), self.threads)
    return (out, ex.astype(bool)) if return_exists else out

  def insert(self, keys, values, clear=False):
    k = self._keys(keys)
    v = self._vals(values, keys)
    self._f["insert"](self._h, k.size, _ptr(k), _ptr(v), int(clear), self.threads)

  def import_values(self, keys, values):
    self.insert(keys, values, clear=True)

  def accum(self, keys, values_or_deltas, exists):
    k = self._keys(keys)
    v = self._vals(values_or_deltas, keys)
    e = np.ascontiguousarray(np.asarray(exists).reshape(-1).astype(np.uint8))
    assert e.size == k.size
    self._f["accum"](self._h, k.size, _ptr(k), _ptr(v), _ptr(e), self.threads)

  def remove(self, keys):
    k = self._keys(keys)
    self._f["remove"](self._h, k.size, _ptr(k))

  def clear(self):
    self._f["clear"](self._h)

  def size(self):
    return int(self._f["size"](self._h))

  def dump(self, offset, length):
    n = self.size()
    cap = max(0, min(length, n - offset)) if offset <= n else 0
    k = np.empty(max(cap, 1), dtype=np.int64)
    v = np.empty((max(cap, 1), self.dim), dtype=self.dtype)
    got = int(self._f["dump"](self._h, _ptr(k), _ptr(v), offset, length))
    return k[:got].copy(), v[:got].copy()

  def export(self):
    return self.dump(0, self.size())

  def export_sorted(self):
    k, v = self.export()
    o = np.argsort(k, kind="stable")
    return k[o], v[o]
