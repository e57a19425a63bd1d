/* TensorFlow custom-op shim, part 2: the DEVICE_GPU kernels of the `TFRA>CuckooHashTable*` ops (ROCm TensorFlow) over
 * libtfra_mi355x.so — what `tfra.dynamic_embedding` creates by default (`CuckooHashTableCreator`) when a Variable is placed
 * on a GPU.
 *
 * Replaces R/.../core/kernels/cuckoo_hashtable_op_gpu.cu.cc (CUDA + the legacy nvhash table, out of scope: SURVEY §2b) in
 * a ROCm build of TFRA's `_cuckoo_hashtable_ops.so`.  The ops themselves stay where they are: the reference's
 * core/ops/cuckoo_hashtable_ops.cc is device-agnostic and also serves the CPU kernels (cuckoo_hashtable_op.cc), so this
 * file registers kernels only — the same 12 names on DEVICE_GPU with the same type constraints as
 * cuckoo_hashtable_op_gpu.cu.cc:698-1058 for K = int64 x V in {float, half, int64, int32, int8} and (int32 keys, float) — round 6: the
 * engine's keys are int64, int32 keys are widened on the device in front of every call, narrowed behind an export and written as 4-byte
 * keys into the key files (mi355x_table_ops.h: Keys64, Export; TFRA_OPTION_KEY_BYTES_ON_DISK).  Behind the kernels: the growing, never-evicting
 * flavour of the table (duplicate keys in one insert: last wins, like the CPU cuckoo table), no dim <= 200 limit, no
 * per-op stream synchronisation, growth in place for big tables.
 *
 * Build: add this file and tf_ops/mi355x_table_ops.h to the `_cuckoo_hashtable_ops.so` target in place of
 * cuckoo_hashtable_op_gpu.cu.cc; link -ltfra_mi355x (see hkv_ops_rocm.cc for the stand-alone command line).
 * tests/test_tf_shim.py checks the registrations against the reference's and compiles this file with -fsyntax-only
 * against tf_ops/stub/.
 */
#include "mi355x_table_ops.h"

namespace tensorflow {
namespace tfra_mi355x {

// registered once, without type constraints (cuckoo_hashtable_op_gpu.cu.cc:698-700,767-769,820-822,865-867,883-885,959-961)
REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableFind").Device(DEVICE_GPU), FindOp);
REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableInsert").Device(DEVICE_GPU), InsertOp<false>);
REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableRemove").Device(DEVICE_GPU), RemoveOp);
REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableSize").Device(DEVICE_GPU), SizeOp);
REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableExport").Device(DEVICE_GPU), ExportOp<true, false>);
REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableImport").Device(DEVICE_GPU), ImportOp);

// per (key, value) type (cuckoo_hashtable_op_gpu.cu.cc:1015-1058)
#define TFRA_REGISTER_CUCKOO_KV(K, V)                                                                                    \
  REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableOfTensors").Device(DEVICE_GPU).TypeConstraint<K>("key_dtype")         \
                              .TypeConstraint<V>("value_dtype"), TableOfTensorsOp<true>);                                \
  REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableClear").Device(DEVICE_GPU).TypeConstraint<K>("key_dtype")             \
                              .TypeConstraint<V>("value_dtype"), ClearOp);                                               \
  REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableAccum").Device(DEVICE_GPU).TypeConstraint<K>("key_dtype")             \
                              .TypeConstraint<V>("value_dtype"), AccumOp<false>);                                        \
  REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableFindWithExists").Device(DEVICE_GPU).TypeConstraint<K>("Tin")          \
                              .TypeConstraint<V>("Tout"), FindWithExistsOp);                                             \
  REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableSaveToFileSystem").Device(DEVICE_GPU)                                 \
                              .TypeConstraint<K>("key_dtype").TypeConstraint<V>("value_dtype")                           \
                              .HostMemory("dirpath").HostMemory("file_name"), SaveToFileSystemOp);                       \
  REGISTER_KERNEL_BUILDER(Name("TFRA>CuckooHashTableLoadFromFileSystem").Device(DEVICE_GPU)                               \
                              .TypeConstraint<K>("key_dtype").TypeConstraint<V>("value_dtype")                           \
                              .HostMemory("dirpath").HostMemory("file_name"), LoadFromFileSystemOp);
#define TFRA_REGISTER_CUCKOO(V) TFRA_REGISTER_CUCKOO_KV(int64_t, V)

TFRA_REGISTER_CUCKOO_KV(int32_t, float);     // cuckoo_hashtable_op_gpu.cu.cc:1058 REGISTER_KERNEL(int32, float)
TFRA_REGISTER_CUCKOO(float);
TFRA_REGISTER_CUCKOO(Eigen::half);
TFRA_REGISTER_CUCKOO(int64_t);
TFRA_REGISTER_CUCKOO(int32_t);
TFRA_REGISTER_CUCKOO(int8_t);

#undef TFRA_REGISTER_CUCKOO
#undef TFRA_REGISTER_CUCKOO_KV

}  // namespace tfra_mi355x
}  // namespace tensorflow
