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
SHIM_FUSED = os.path.join(ROOT, "tf_ops", "fused_ops_rocm.cc")
GOLDEN_FUSED = os.path.join(ROOT, "tests", "golden", "fused_op_surface.json")
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
  for src in (SHIM, SHIM_CUCKOO, SHIM_FUSED):
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
  assert pairs == want["gpu_type_pairs"]                         # every (key, value) pair of the reference, (int32, float) included
  assert ["int32_t", "float"] in pairs                           # round 6: int32 keys, widened in front of the int64 engine
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


def test_fused_shim_registers_the_fused_paths_as_ops_with_gpu_kernels():
  """tf_ops/fused_ops_rocm.cc: what north_star names next to the table ops — the sparse embedding_lookup gather and the optimizer
  scatter-update — plus the overlapped step and the multi-GPU route, each ONE registered op over the C ABI.  Surface pinned in
  tests/golden/fused_op_surface.json (tests/golden/make_op_surface.py); every op has its DEVICE_GPU kernel; nothing collides with the
  reference's ops."""
  mod = _extractor()
  text = open(SHIM_FUSED).read()
  want = json.load(open(GOLDEN_FUSED))
  got = mod.parse_register_ops(text)
  assert got == want["ops"]
  kernels = mod.parse_gpu_registrations_detailed(text)
  assert kernels == want["gpu_kernels"] and set(kernels) == set(got)
  names = set(got)
  for o in ("Sgd", "Adam", "Adagrad", "Ftrl"):
    assert "TFRA>HkvHashTableApplySparse" + o in names and "TFRA>RouteApply" + o in names
  assert {"TFRA>HkvHashTableOfTensorsWithSlots", "TFRA>HkvHashTableEmbeddingLookup", "TFRA>HkvHashTableInsertN", "TFRA>HkvHashTableLookupAssignStep",
          "TFRA>HkvHashTableLookupAssignFlush", "TFRA>RcclUniqueId", "TFRA>RouteCreate", "TFRA>RouteFeed", "TFRA>RouteLookup",
          "TFRA>AssignRouteCreate", "TFRA>AssignRouteFeed", "TFRA>AssignRouteStep", "TFRA>AssignRouteFlush"} <= names
  # the routed assign step moves rows of the table's own value type (any of the GPU value types), one Step = lookup + the previous write-back
  assert got["TFRA>AssignRouteStep"]["inputs"] == ["route_handle: resource", "default_value: value_dtype", "prev_values: value_dtype"]
  assert got["TFRA>AssignRouteStep"]["outputs"] == ["values: value_dtype"] and kernels["TFRA>AssignRouteStep"] == ["value_dtype"]
  ref = set(json.load(open(GOLDEN))["ops"]) | set(json.load(open(GOLDEN_CUCKOO))["ops"])
  assert not (names & ref)                                         # loaded next to the reference's ops: no name registered twice
  # the creator with slots = the reference creator's attrs + the two new ones, so TFRA's Python wrapper passes the same kwargs
  base = json.load(open(GOLDEN))["ops"]["TFRA>HkvHashTableOfTensors"]
  slots = got["TFRA>HkvHashTableOfTensorsWithSlots"]
  assert set(base["attrs"]) < set(slots["attrs"]) and slots["outputs"] == base["outputs"] and slots["stateful"]
  assert sorted(set(slots["attrs"]) - set(base["attrs"])) == ["optimizer_slots: int >= 0 = 0", "slot_init: list(float) = []"]
  # the lookup op returns everything embedding_lookup_unique needs without a host read: rows, unique ids, inverse index, count
  assert got["TFRA>HkvHashTableEmbeddingLookup"]["outputs"] == ["values: value_dtype", "unique_ids: key_dtype", "idx: int32", "num_unique: int64"]
  # hyper-parameters are host-memory scalar inputs (a schedule is a tensor), like TensorFlow's ResourceSparseApply* ops
  assert got["TFRA>HkvHashTableApplySparseAdam"]["inputs"][4:] == ["lr: float", "beta1_power: float", "beta2_power: float", "beta1: float",
                                                                   "beta2: float", "epsilon: float"]
  for name in names:
    if "ApplySparse" in name or "RouteApply" in name:
      m = __import__("re").search(r'Name\("%s"\)((?:[\s\\]*\.\w+(?:<[^>]*>)?\([^()]*\))*)' % name.replace(">", ">"), text)
      assert m and '.HostMemory("lr")' in m.group(1), name


def test_fused_shim_binds_only_declared_abi_entry_points_and_reaches_the_fused_calls():
  import re
  hdr = open(os.path.join(ROOT, "include", "tfra_mi355x.h")).read()
  declared = set(re.findall(r"\b(tfra_\w+)\s*\(", hdr))
  used = set(re.findall(r"\b(tfra_[a-z_0-9]+)\(", open(SHIM_FUSED).read() + open(SHIM_COMMON).read()))
  used -= {"tfra_mi355x"}
  assert used <= declared, sorted(used - declared)
  for fn in ("tfra_table_apply_sparse", "tfra_table_find_unique", "tfra_table_find", "tfra_table_insert_or_assign_n",
             "tfra_step_driver_create", "tfra_table_step_overlap", "tfra_table_step_overlap_flush", "tfra_workspace_create",
             "tfra_rccl_unique_id", "tfra_rccl_transport_create", "tfra_route_create", "tfra_route_feed", "tfra_route_lookup", "tfra_route_apply",
             "tfra_assign_route_create", "tfra_assign_route_feed", "tfra_assign_route_step", "tfra_assign_route_flush"):
    assert fn in used, fn


def test_tfra_side_binding_calls_registered_ops_with_their_signatures():
  """tf_ops/tfra_fused_binding.py (the Python a TFRA maintainer adds: INTEGRATION.md §2.1) is valid Python and every generated-wrapper
  call in it — `ops.tfra_<snake_case_op>(inputs..., attr=...)` — names a registered fused op, passes exactly its inputs positionally
  and only its attrs by keyword."""
  import ast
  import re
  mod = _extractor()
  ops = mod.parse_register_ops(open(SHIM_FUSED).read())
  snake = {re.sub(r"(?<!^)(?=[A-Z])", "_", n.split(">")[1]).lower(): (n, d) for n, d in ops.items()}
  src = open(os.path.join(ROOT, "tf_ops", "tfra_fused_binding.py")).read()
  tree = ast.parse(src)
  seen = set()
  for node in ast.walk(tree):
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("tfra_"):
      key = node.func.attr[len("tfra_"):]
      assert key in snake, node.func.attr
      name, d = snake[key]
      seen.add(name)
      assert len(node.args) == len(d["inputs"]), (name, len(node.args), d["inputs"])
      attrs = {a.split(":")[0].strip() for a in d["attrs"]} | {"name"}
      for kw in node.keywords:
        assert kw.arg in attrs, (name, kw.arg)
  # every family of fused ops is bound from Python
  for want in ("TFRA>HkvHashTableOfTensorsWithSlots", "TFRA>HkvHashTableEmbeddingLookup", "TFRA>HkvHashTableApplySparseAdam",
               "TFRA>HkvHashTableApplySparseSgd", "TFRA>HkvHashTableApplySparseAdagrad", "TFRA>HkvHashTableApplySparseFtrl",
               "TFRA>HkvHashTableLookupAssignStep", "TFRA>HkvHashTableLookupAssignFlush", "TFRA>RcclUniqueId", "TFRA>RouteCreate",
               "TFRA>RouteFeed", "TFRA>RouteLookup", "TFRA>RouteApplyAdam", "TFRA>AssignRouteCreate", "TFRA>AssignRouteFeed",
               "TFRA>AssignRouteStep", "TFRA>AssignRouteFlush"):
    assert want in seen, want
