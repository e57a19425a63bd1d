"""Replays a golden op sequence (tests/golden/make_golden.py) on any table factory and checks
every recorded reference output bit-exactly."""
import numpy as np

KIND = {0: "insert", 1: "find", 2: "accum", 3: "remove", 4: "import", 5: "clear"}


def replay(path, make):
  z = np.load(path)
  dim = int(z["dim"]); dtype = np.dtype(str(z["dtype"]))
  t = make(dim, dtype)
  for s in range(int(z["n_ops"])):
    p = "op%03d_" % s
    kind = KIND[int(z[p + "kind"])]
    if kind == "insert":
      t.insert(z[p + "keys"], z[p + "vals"])
    elif kind == "find":
      v, e = t.find(z[p + "keys"], z[p + "vals"], True)
      np.testing.assert_array_equal(e, z[p + "exp_exists"], err_msg="%s %s" % (path, p))
      np.testing.assert_array_equal(v.view(np.uint8), z[p + "exp_vals"].view(np.uint8), err_msg="%s %s" % (path, p))
    elif kind == "accum":
      t.accum(z[p + "keys"], z[p + "vals"], z[p + "exists"])
    elif kind == "remove":
      t.remove(z[p + "keys"])
    elif kind == "import":
      t.import_values(z[p + "keys"], z[p + "vals"])
    else:
      t.clear()
    assert t.size() == int(z[p + "size"]), "%s %s size" % (path, p)
  k, v = t.export_sorted()
  np.testing.assert_array_equal(k, z["final_keys"])
  np.testing.assert_array_equal(v.view(np.uint8), z["final_vals"].view(np.uint8))
