"""Generates tests/golden/ref_ops_*.npz by running the REAL reference engine
(oracle/_ref/libtfra_ref.so = /root/reference's cuckoohash_map.hh compiled in place) on seeded op
sequences.  Run in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

Each fixture stores the op list (inputs) and, for every find, the expected values/exists, the
table size after every op, and the final sorted export — so any engine can be replayed against
the reference WITHOUT the reference being present (tests/golden/replay.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OPS = {"insert": 0, "find": 1, "accum": 2, "remove": 3, "import": 4, "clear": 5}


def make(seed, dim, dtype, steps, max_n, universe_size, path):
  rng = np.random.default_rng(seed)
  t = oracle.CpuTable(dim, dtype, kind="reference")
  i64 = np.iinfo(np.int64)
  universe = rng.integers(-2**62, 2**62, size=universe_size, dtype=np.int64)
  universe[:5] = [0, -1, i64.min, i64.min + 1, i64.max]
  out = {"dim": np.int64(dim), "dtype": np.array(np.dtype(dtype).str), "n_ops": np.int64(steps)}

  def vals(n):
    if np.issubdtype(np.dtype(dtype), np.floating):
      return (rng.standard_normal((n, dim)) * 3).astype(dtype)
    return rng.integers(-1000, 1000, size=(n, dim)).astype(dtype)

  for s in range(steps):
    n = int(rng.integers(1, max_n))
    keys = rng.choice(universe, size=n)
    r = rng.random()
    p = "op%03d_" % s
    if r < 0.30:
      v = vals(n); t.insert(keys, v)
      out[p + "kind"] = np.int64(OPS["insert"]); out[p + "keys"] = keys; out[p + "vals"] = v
    elif r < 0.60:
      full = rng.random() < 0.5
      d = vals(n) if full else vals(1)[0]
      fv, fe = t.find(keys, d, True)
      out[p + "kind"] = np.int64(OPS["find"]); out[p + "keys"] = keys; out[p + "vals"] = d
      out[p + "exp_vals"] = fv; out[p + "exp_exists"] = fe
    elif r < 0.80:
      v = vals(n); ex = rng.random(n) < 0.5; t.accum(keys, v, ex)
      out[p + "kind"] = np.int64(OPS["accum"]); out[p + "keys"] = keys; out[p + "vals"] = v; out[p + "exists"] = ex
    elif r < 0.93:
      t.remove(keys)
      out[p + "kind"] = np.int64(OPS["remove"]); out[p + "keys"] = keys
    elif r < 0.98:
      keys = np.unique(keys); v = vals(keys.size); t.import_values(keys, v)
      out[p + "kind"] = np.int64(OPS["import"]); out[p + "keys"] = keys; out[p + "vals"] = v
    else:
      t.clear()
      out[p + "kind"] = np.int64(OPS["clear"])
    out[p + "size"] = np.int64(t.size())
  k, v = t.export_sorted()
  out["final_keys"] = k; out["final_vals"] = v
  np.savez_compressed(path, **out)
  print(path, os.path.getsize(path), "bytes; final size", k.size)


if __name__ == "__main__":
  here = os.path.dirname(os.path.abspath(__file__))
  oracle.build()
  make(20250205, 16, np.float32, 40, 400, 900, os.path.join(here, "ref_ops_f32_d16.npz"))
  make(20250206, 64, np.float32, 24, 200, 500, os.path.join(here, "ref_ops_f32_d64.npz"))
  make(20250207, 3, np.int32, 40, 300, 600, os.path.join(here, "ref_ops_i32_d3.npz"))
  make(20250218, 130, np.float32, 16, 120, 300, os.path.join(here, "ref_ops_f32_d130.npz"))
  make(20250209, 8, np.int64, 30, 300, 600, os.path.join(here, "ref_ops_i64_d8.npz"))
