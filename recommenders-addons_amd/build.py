"""Builds the in-tree HIP shared library for gfx950 (MI355X).

  python recommenders-addons_amd/build.py [--force]

Output: recommenders-addons_amd/tfra_amd/lib/libtfra_mi355x.so  (git-ignored; travels to the GPU
box with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "tfra_amd", "lib")
LIB = os.path.join(LIBDIR, "libtfra_mi355x.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3"] + os.environ.get("TFRA_EXTRA_FLAGS", "").split() + ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-variable", "-Wno-unused-function", "-Wno-unused-result"]


def _deps():
  hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
  hdrs.append(os.path.join(HERE, "..", "include", "tfra_mi355x.h"))
  return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, newest_hdr):
  obj = os.path.join(OBJ, os.path.basename(src) + ".o")
  if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
      and os.path.getmtime(obj) > newest_hdr):
    return obj, False
  extra = ["-DTFRA_WITH_TUNING"] if os.environ.get("TFRA_WITH_TUNING") else []
  cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
  subprocess.check_call(cmd)
  return obj, True


def build(force=False, verbose=True):
  os.makedirs(OBJ, exist_ok=True)
  os.makedirs(LIBDIR, exist_ok=True)
  srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
  if os.environ.get("TFRA_WITH_TUNING"):  # kernel-ablation entry points for scripts/microbench_find.py
    tdir = os.path.join(CSRC, "tuning")
    srcs += sorted(os.path.join(tdir, f) for f in os.listdir(tdir) if f.endswith(".hip"))
  newest = _deps()
  with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
    res = list(ex.map(lambda s: _compile(s, force, newest), srcs))
  objs = [o for o, _ in res]
  if force or any(ch for _, ch in res) or not os.path.exists(LIB):
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB])
    if verbose:
      print("built", LIB)
  return LIB


if __name__ == "__main__":
  build(force="--force" in sys.argv)
