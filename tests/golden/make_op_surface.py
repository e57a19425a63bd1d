"""Extracts the op surface (names, inputs, outputs, attrs, statefulness) of the reference's `TFRA>HkvHashTable*` ops from
R/.../core/ops/hkv_hashtable_ops.cc and the GPU kernel registrations from R/.../core/kernels/hkv_hashtable_op_gpu.cu.cc
into tests/golden/hkv_op_surface.json, and the same for the `TFRA>CuckooHashTable*` ops (core/ops/cuckoo_hashtable_ops.cc,
core/kernels/cuckoo_hashtable_op_gpu.cu.cc) into tests/golden/cuckoo_op_surface.json (the reference tree is absent on the
GPU box and in CI images).

  python tests/golden/make_op_surface.py
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tensorflow_recommenders_addons/dynamic_embedding/core"


def parse_register_ops(text, prefix_macro=True):
  """-> {op name: {inputs, outputs, attrs, stateful}} from REGISTER_OP(...) builder chains."""
  ops = {}
  for m in re.finditer(r"REGISTER_OP\(\s*([^)]*?\)?)\s*\)\s*((?:\s*\.\w+\((?:[^()]|\([^()]*\))*\))*)", text):
    raw = m.group(1).strip()
    mm = re.match(r'PREFIX_OP_NAME\((\w+)\)', raw)
    name = ("TFRA>" + mm.group(1)) if mm else raw.strip('"')
    chain = m.group(2)
    d = {"inputs": re.findall(r'\.Input\("([^"]*)"\)', chain), "outputs": re.findall(r'\.Output\("([^"]*)"\)', chain),
         "attrs": sorted(re.findall(r'\.Attr\("([^"]*)"\)', chain)), "stateful": ".SetIsStateful()" in chain}
    ops[name] = d
  return ops


def parse_gpu_registrations(text):
  """-> (sorted op names registered on DEVICE_GPU, sorted value types of the per-type registration macro)."""
  names = set()
  for m in re.finditer(r'Name\(\s*(?:PREFIX_OP_NAME\((\w+)\)|"([^"]+)")\s*\)[\s\\]*\.Device\(DEVICE_GPU\)', text):
    names.add("TFRA>" + m.group(1) if m.group(1) else m.group(2))
  types = set()
  for m in re.finditer(r"^\s*(?:REGISTER_HKV_TABLE\(\s*int64\s*,|TFRA_REGISTER_HKV\()\s*([\w:]+)\s*\)\s*;", text, re.M):
    types.add({"int8": "int8_t", "int32": "int32_t", "int64": "int64_t"}.get(m.group(1), m.group(1)))
  return sorted(names), sorted(types)


def parse_cuckoo_gpu_types(text):
  """-> sorted [key type, value type] pairs of the per-type registration macro of the cuckoo GPU kernels
  (`REGISTER_KERNEL(int64, float);` in the reference, `TFRA_REGISTER_CUCKOO(float);` = int64 keys in the shim)."""
  norm = {"int8": "int8_t", "int32": "int32_t", "int64": "int64_t"}
  pairs = set()
  for m in re.finditer(r"^\s*REGISTER_KERNEL\(\s*([\w:]+)\s*,\s*([\w:]+)\s*\)\s*;", text, re.M):
    pairs.add((norm.get(m.group(1), m.group(1)), norm.get(m.group(2), m.group(2))))
  for m in re.finditer(r"^\s*TFRA_REGISTER_CUCKOO\(\s*([\w:]+)\s*\)\s*;", text, re.M):
    pairs.add(("int64_t", norm.get(m.group(1), m.group(1))))
  for m in re.finditer(r"^\s*TFRA_REGISTER_CUCKOO_KV\(\s*([\w:]+)\s*,\s*([\w:]+)\s*\)\s*;", text, re.M):
    pairs.add((norm.get(m.group(1), m.group(1)), norm.get(m.group(2), m.group(2))))
  return sorted(list(p) for p in pairs)


def parse_gpu_registrations_detailed(text):
  """-> {op name: sorted type-constraint attr names ([] = registered once, unconstrained)} for DEVICE_GPU registrations."""
  out = {}
  for m in re.finditer(r'Name\(\s*(?:PREFIX_OP_NAME\((\w+)\)|"([^"]+)")\s*\)((?:[\s\\]*\.\w+(?:<[^>]*>)?\([^()]*\))*)', text):
    chain = m.group(3)
    if ".Device(DEVICE_GPU)" not in chain:
      continue
    name = "TFRA>" + m.group(1) if m.group(1) else m.group(2)
    out[name] = sorted(re.findall(r'\.TypeConstraint<[^>]*>\("(\w+)"\)', chain))
  return out


def main():
  ops = parse_register_ops(open(os.path.join(REF, "ops", "hkv_hashtable_ops.cc")).read())
  names, types = parse_gpu_registrations(open(os.path.join(REF, "kernels", "hkv_hashtable_op_gpu.cu.cc")).read())
  out = {"source": ["core/ops/hkv_hashtable_ops.cc:133-339", "core/kernels/hkv_hashtable_op_gpu.cu.cc:809-811,1058-1138"],
         "ops": ops, "gpu_kernels": names, "gpu_value_types": types}
  with open(os.path.join(HERE, "hkv_op_surface.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print(len(ops), "ops;", len(names), "GPU kernels;", types)
  cops = parse_register_ops(open(os.path.join(REF, "ops", "cuckoo_hashtable_ops.cc")).read())
  ctext = open(os.path.join(REF, "kernels", "cuckoo_hashtable_op_gpu.cu.cc")).read()
  out = {"source": ["core/ops/cuckoo_hashtable_ops.cc:134-310", "core/kernels/cuckoo_hashtable_op_gpu.cu.cc:698-1058"],
         "ops": cops, "gpu_kernels": parse_gpu_registrations_detailed(ctext), "gpu_type_pairs": parse_cuckoo_gpu_types(ctext)}
  with open(os.path.join(HERE, "cuckoo_op_surface.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print(len(cops), "cuckoo ops;", len(out["gpu_kernels"]), "GPU kernels;", out["gpu_type_pairs"])


def main_fused():
  """The fused ops are this repository's own (no reference text to compare with): their surface is pinned as it is registered in
  tf_ops/fused_ops_rocm.cc, so that a change of an input order / attr default shows up as a diff of the golden file."""
  root = os.path.dirname(os.path.dirname(HERE))
  text = open(os.path.join(root, "tf_ops", "fused_ops_rocm.cc")).read()
  out = {"source": ["tf_ops/fused_ops_rocm.cc"], "ops": parse_register_ops(text), "gpu_kernels": parse_gpu_registrations_detailed(text),
         "replaces": {
             "TFRA>HkvHashTableEmbeddingLookup": "python/ops/dynamic_embedding_ops.py:99-117 (tf.unique + embedding_lookup + tf.gather)",
             "TFRA>HkvHashTableApplySparse*": "python/ops/dynamic_embedding_optimizer.py:165-204 (_resource_apply_sparse_duplicate_indices + (1+S) finds / upserts), slots :870-958",
             "TFRA>HkvHashTableLookupAssignStep": "core/kernels/hkv_hashtable_op_gpu.cu.cc:182-290 (Find + Insert in order)",
             "TFRA>Route*": "python/ops/shadow_embedding_ops.py:397-447 (HvdAllToAllEmbedding's alltoall route)"}}
  with open(os.path.join(HERE, "fused_op_surface.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print(len(out["ops"]), "fused ops;", len(out["gpu_kernels"]), "GPU kernels")


if __name__ == "__main__":
  if os.path.isdir(REF):
    main()
  main_fused()
