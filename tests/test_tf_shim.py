"""R1: the TensorFlow custom-op shim tf_ops/hkv_ops_rocm.cc registers exactly the reference's
`TFRA>HkvHashTable*` op surface — names, inputs, outputs, attrs (with defaults and constraints), statefulness — and the
same DEVICE_GPU kernels for the same value types, and it is valid C++ against the TensorFlow API it names.

Reference: R/.../core/ops/hkv_hashtable_ops.cc:133-339, R/.../core/kernels/hkv_hashtable_op_gpu.cu.cc:809-811,1058-1138
(extracted into tests/golden/hkv_op_surface.json by tests/golden/make_op_surface.py)."""
import importlib.util
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tf_ops", "hkv_ops_rocm.cc")
SHIM_CUCKOO = os.path.join(ROOT, "tf_ops", "cuckoo_ops_rocm.cc")
SHIM_COMMON = os.path.join(ROOT, "tf_ops", "mi355x_table_ops.h")
GOLDEN = os.path.join(ROOT, "tests", "golden", "hkv_op_surface.json")
GOLDEN_CUCKOO = os.path.join(ROOT, "tests", "golden", "cuckoo_op_surface.json")


def _extractor():
  spec = importlib.util.spec_from_file_location("make_op_surface", os.path.join(ROOT, "tests", "golden", "make_op_surface.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_shim_registers_the_reference_op_surface():
  mod = _extractor()
  want = json.load(open(GOLDEN))
  got = mod.parse_register_ops(open(SHIM).read())
  assert sorted(got) == sorted(want["ops"]), (sorted(set(want["ops"]) - set(got)), sorted(set(got) - set(want["ops"])))
  assert len(got) == 14
  for name, ref in want["ops"].items():
    assert got[name]["inputs"] == ref["inputs"], name      # order matters: it is the op's positional signature
    assert got[name]["outputs"] == ref["outputs"], name
    assert got[name]["attrs"] == ref["attrs"], name        # incl. defaults ("init_capacity: int = 0") and constraints (">= 1")
    assert got[name]["stateful"] == ref["stateful"], name


def test_shim_registers_the_reference_gpu_kernels():
  mod = _extractor()
  want = json.load(open(GOLDEN))
  names, types = mod.parse_gpu_registrations(open(SHIM).read())
  assert names == want["gpu_kernels"]
  assert len(names) == 14
  assert types == want["gpu_value_types"] == ["Eigen::half", "bfloat16", "float", "int32_t", "int64_t", "int8_t"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_golden_surface_is_what_the_reference_registers_now():
  mod = _extractor()
  want = json.load(open(GOLDEN))
  ops = mod.parse_register_ops(open(os.path.join(mod.REF, "ops", "hkv_hashtable_ops.cc")).read())
  names, types = mod.parse_gpu_registrations(open(os.path.join(mod.REF, "kernels", "hkv_hashtable_op_gpu.cu.cc")).read())
  assert ops == want["ops"] and names == want["gpu_kernels"] and types == want["gpu_value_types"]


def test_shim_binds_only_declared_abi_entry_points():
  import re
  hdr = open(os.path.join(ROOT, "include", "tfra_mi355x.h")).read()
  declared = set(re.findall(r"\b(tfra_\w+)\s*\(", hdr))
  used = set(re.findall(r"\b(tfra_(?:table|last)\w+)\s*\(", open(SHIM).read() + open(SHIM_COMMON).read() + open(SHIM_CUCKOO).read()))
  assert used and used <= declared, sorted(used - declared)
  # every engine call of the reference adapter (lookup_table_op_hkv.h:515-756) is reachable from an op kernel
  for fn in ("tfra_table_create", "tfra_table_destroy", "tfra_table_find", "tfra_table_insert_or_assign", "tfra_table_accum_or_assign",
             "tfra_table_erase", "tfra_table_clear", "tfra_table_size", "tfra_table_size_to_device", "tfra_table_capacity",
             "tfra_table_export_batch", "tfra_table_save", "tfra_table_load"):
    assert fn in used, fn


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the ROCm headers")
def test_shim_is_valid_cxx_against_the_tensorflow_api_it_uses():
  """-fsyntax-only against tf_ops/stub/ (declarations of the TensorFlow 2.16 API the shim names) + the real C ABI header."""
  for src in (SHIM, SHIM_CUCKOO):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "tf_ops", "stub"),
           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, (src, r.stderr[-3000:])


def test_cuckoo_shim_registers_the_reference_gpu_kernels():
  """`TFRA>CuckooHashTable*` (what `CuckooHashTableCreator`, TFRA's default, instantiates): the ops stay in the reference's
  device-agnostic cuckoo_hashtable_ops.cc, the shim replaces cuckoo_hashtable_op_gpu.cu.cc — the same 12 kernels on
  DEVICE_GPU with the same type-constraint attrs, for the reference's (int64, V) pairs."""
  mod = _extractor()
  want = json.load(open(GOLDEN_CUCKOO))
  text = open(SHIM_CUCKOO).read()
  assert mod.parse_register_ops(text) == {}                      # kernels only: registering the ops twice would not load
  got = mod.parse_gpu_registrations_detailed(text)
  assert got == want["gpu_kernels"] and len(got) == 12
  assert set(got) == set(want["ops"])                            # every cuckoo op has its GPU kernel
  pairs = mod.parse_cuckoo_gpu_types(text)
  assert pairs == [p for p in want["gpu_type_pairs"] if p[0] == "int64_t"]
  assert [p for p in want["gpu_type_pairs"] if p[0] != "int64_t"] == [["int32_t", "float"]]   # the one pair not covered (int32 keys)
  # the shared op kernels read their inputs by position: the cuckoo ops are the Hkv ops without the trailing `scores`
  hkv = json.load(open(GOLDEN))["ops"]
  for name, ref in want["ops"].items():
    twin = hkv[name.replace("Cuckoo", "Hkv")]
    assert len(ref["outputs"]) == len(twin["outputs"]), name
    tin = [i.split(":")[0] for i in twin["inputs"] if not i.startswith("scores")]
    assert [i.split(":")[0] for i in ref["inputs"]] == tin, name


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_cuckoo_golden_surface_is_what_the_reference_registers_now():
  mod = _extractor()
  want = json.load(open(GOLDEN_CUCKOO))
  ops = mod.parse_register_ops(open(os.path.join(mod.REF, "ops", "cuckoo_hashtable_ops.cc")).read())
  text = open(os.path.join(mod.REF, "kernels", "cuckoo_hashtable_op_gpu.cu.cc")).read()
  assert ops == want["ops"] and mod.parse_gpu_registrations_detailed(text) == want["gpu_kernels"]
  assert mod.parse_cuckoo_gpu_types(text) == want["gpu_type_pairs"]
