/* TensorFlow custom-op shim: the `TFRA>HkvHashTable*` ops on DEVICE_GPU (ROCm TensorFlow) over libtfra_mi355x.so.
 *
 * Drop-in for TFRA's `_hkv_ops.so` (R/tensorflow_recommenders_addons/dynamic_embedding/core/BUILD:121-146): the same
 * 14 op names, inputs, outputs and attrs as R/.../core/ops/hkv_hashtable_ops.cc:133-339, the same kernels registered
 * for K = int64 x V in {float, int8, int32, int64, half, bfloat16} as R/.../core/kernels/hkv_hashtable_op_gpu.cu.cc:
 * 1133-1138, so `tf.load_op_library` + TFRA's generated Python wrappers (PY/hkv_hashtable_ops.py) work unchanged.
 * What differs from the reference kernels is only what sits behind them: one runtime-typed table class over the C ABI
 * (include/tfra_mi355x.h) instead of a template per (K, V) over HierarchicalKV, no per-op stream synchronisation (every
 * ABI entry point is stream-ordered), no temporary device buffers, and the default fill fused into find.
 *
 * Build (on a box with ROCm TensorFlow):
 *   hipcc -std=c++17 -shared -fPIC tf_ops/hkv_ops_rocm.cc -o _hkv_ops.so -Iinclude \
 *         $(python -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_compile_flags()))') \
 *         $(python -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_link_flags()))') \
 *         -Lrecommenders-addons_amd/tfra_amd/lib -ltfra_mi355x
 * TensorFlow is not installed in this repository's build image: tests/test_tf_shim.py checks the op surface against
 * the reference's REGISTER_OP text and compiles this file with -fsyntax-only against the declarations in tf_ops/stub/
 * (a few hundred lines naming exactly the TensorFlow API used here).
 */
#include "mi355x_table_ops.h"

namespace tensorflow {
namespace tfra_mi355x {

using shape_inference::DimensionHandle;
using shape_inference::InferenceContext;
using shape_inference::ShapeAndType;
using shape_inference::ShapeHandle;

// =========================================== op registrations ===========================================
// Shape functions: what the reference's do (hkv_hashtable_ops.cc:47-131), written once.

static Status ScalarHandle(InferenceContext* c) {
  ShapeHandle h;
  return c->WithRank(c->input(0), 0, &h);
}

// value shape of a lookup = keys.shape + value_shape recorded on the resource handle (when known)
static Status LookupShape(InferenceContext* c, ShapeHandle keys, bool is_lookup, ShapeHandle* out) {
  const std::vector<ShapeAndType>* hd = c->input_handle_shapes_and_types(0);
  if (hd == nullptr || hd->size() != 2) {
    *out = c->UnknownShape();
    return OkStatus();
  }
  DataType kt, vt;
  TF_RETURN_IF_ERROR(c->GetAttr("key_dtype", &kt));
  TF_RETURN_IF_ERROR(c->GetAttr("value_dtype", &vt));
  if ((*hd)[0].dtype != kt || (*hd)[1].dtype != vt)
    return errors::InvalidArgument("Trying to read value with wrong dtype. Expected ", DataTypeString((*hd)[0].dtype), "-",
                                   DataTypeString((*hd)[1].dtype), " got ", DataTypeString(kt), "-", DataTypeString(vt));
  ShapeHandle value = (*hd)[1].shape;
  if (is_lookup) {
    if (c->RankKnown(keys) && c->RankKnown((*hd)[0].shape)) {
      ShapeHandle prefix;
      TF_RETURN_IF_ERROR(c->Subshape(keys, 0, c->Rank(keys) - c->Rank((*hd)[0].shape), &prefix));
      TF_RETURN_IF_ERROR(c->Concatenate(prefix, value, out));
    } else {
      *out = c->UnknownShape();
    }
  } else {
    TF_RETURN_IF_ERROR(c->Concatenate(keys, value, out));
  }
  return OkStatus();
}

REGISTER_OP("TFRA>HkvHashTableOfTensors")
    .Output("table_handle: resource")
    .Attr("container: string = ''")
    .Attr("shared_name: string = ''")
    .Attr("use_node_name_sharing: bool = false")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .Attr("value_shape: shape = {}")
    .Attr("init_capacity: int = 0")
    .Attr("max_capacity: int = 0")
    .Attr("max_hbm_for_vectors: int = 0")
    .Attr("step_per_epoch: int = 0")
    .Attr("strategy: int = 0")
    .Attr("reserved_key_start_bit: int = 0")
    .SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      PartialTensorShape vp;
      TF_RETURN_IF_ERROR(c->GetAttr("value_shape", &vp));
      ShapeHandle vs;
      TF_RETURN_IF_ERROR(c->MakeShapeFromPartialTensorShape(vp, &vs));
      DataType kt, vt;
      TF_RETURN_IF_ERROR(c->GetAttr("key_dtype", &kt));
      TF_RETURN_IF_ERROR(c->GetAttr("value_dtype", &vt));
      c->set_output(0, c->Scalar());
      c->set_output_handle_shapes_and_types(0, std::vector<ShapeAndType>{{c->Scalar(), kt}, {vs, vt}});
      return OkStatus();
    });

REGISTER_OP("TFRA>HkvHashTableFind")
    .Input("table_handle: resource")
    .Input("keys: key_dtype")
    .Input("default_value: value_dtype")
    .Output("values: value_dtype")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(ScalarHandle(c));
      ShapeHandle out;
      TF_RETURN_IF_ERROR(LookupShape(c, c->input(1), true, &out));
      c->set_output(0, out);
      return OkStatus();
    });

REGISTER_OP("TFRA>HkvHashTableFindWithExists")
    .Input("table_handle: resource")
    .Input("keys: key_dtype")
    .Input("default_value: value_dtype")
    .Output("values: value_dtype")
    .Output("exists: bool")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(ScalarHandle(c));
      ShapeHandle out;
      TF_RETURN_IF_ERROR(LookupShape(c, c->input(1), true, &out));
      c->set_output(0, out);
      c->set_output(1, c->UnknownShapeOfRank(1));
      return OkStatus();
    });

REGISTER_OP("TFRA>HkvHashTableInsert")
    .Input("table_handle: resource")
    .Input("keys: key_dtype")
    .Input("values: value_dtype")
    .Input("scores: int64")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .SetShapeFn(ScalarHandle);

REGISTER_OP("TFRA>HkvHashTableAccum")
    .Input("table_handle: resource")
    .Input("keys: key_dtype")
    .Input("values_or_deltas: value_dtype")
    .Input("exists: bool")
    .Input("scores: int64")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .SetShapeFn(ScalarHandle);

REGISTER_OP("TFRA>HkvHashTableRemove")
    .Input("table_handle: resource")
    .Input("keys: key_dtype")
    .Attr("key_dtype: type")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(ScalarHandle(c));
      ShapeHandle k;
      return c->WithRankAtLeast(c->input(1), 1, &k);
    });

REGISTER_OP("TFRA>HkvHashTableClear")
    .Input("table_handle: resource")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type");

REGISTER_OP("TFRA>HkvHashTableSize")
    .Input("table_handle: resource")
    .Output("size: int64")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(ScalarHandle(c));
      c->set_output(0, c->Scalar());
      return OkStatus();
    });

REGISTER_OP("TFRA>HkvHashTableExport")
    .Input("table_handle: resource")
    .Output("keys: key_dtype")
    .Output("values: value_dtype")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(ScalarHandle(c));
      ShapeHandle keys = c->UnknownShapeOfRank(1), values;
      TF_RETURN_IF_ERROR(LookupShape(c, keys, false, &values));
      c->set_output(0, keys);
      c->set_output(1, values);
      return OkStatus();
    });

REGISTER_OP("TFRA>HkvHashTableExportWithScores")
    .Input("table_handle: resource")
    .Output("keys: key_dtype")
    .Output("values: value_dtype")
    .Output("scores: int64")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .Attr("split_size: int")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(ScalarHandle(c));
      for (int i = 0; i < 3; ++i) c->set_output(i, c->UnknownShapeOfRank(1));
      return OkStatus();
    });

REGISTER_OP("TFRA>HkvHashTableExportKeysAndScores")
    .Input("table_handle: resource")
    .Output("keys: key_dtype")
    .Output("scores: int64")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .Attr("split_size: int")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(ScalarHandle(c));
      for (int i = 0; i < 2; ++i) c->set_output(i, c->UnknownShapeOfRank(1));
      return OkStatus();
    });

REGISTER_OP("TFRA>HkvHashTableImport")
    .Input("table_handle: resource")
    .Input("keys: key_dtype")
    .Input("values: value_dtype")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(ScalarHandle(c));
      ShapeHandle keys;
      TF_RETURN_IF_ERROR(c->WithRank(c->input(1), 1, &keys));
      DimensionHandle n;
      return c->Merge(c->Dim(keys, 0), c->Dim(c->input(2), 0), &n);
    });

REGISTER_OP("TFRA>HkvHashTableSaveToFileSystem")
    .Input("table_handle: resource")
    .Input("dirpath: string")
    .Input("file_name: string")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .Attr("dirpath_env: string")
    .Attr("append_to_file: bool")
    .Attr("buffer_size: int >= 1");

REGISTER_OP("TFRA>HkvHashTableLoadFromFileSystem")
    .Input("table_handle: resource")
    .Input("dirpath: string")
    .Input("file_name: string")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .Attr("dirpath_env: string")
    .Attr("load_entire_dir: bool")
    .Attr("buffer_size: int >= 1");

// =========================================== kernel registrations ===========================================
// K = int64 x V in {float, int8, int32, int64, half, bfloat16} on DEVICE_GPU (= the ROCm device string too), as
// hkv_hashtable_op_gpu.cu.cc:1058-1138; Remove is registered once without type constraints (:809-811).

REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableRemove").Device(DEVICE_GPU), RemoveOp);

#define TFRA_REGISTER_HKV(V)                                                                                            \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableOfTensors").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")     \
                              .TypeConstraint<V>("value_dtype"), TableOfTensorsOp<false>);                                     \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableFind").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")          \
                              .TypeConstraint<V>("value_dtype"), FindOp);                                               \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableFindWithExists").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")\
                              .TypeConstraint<V>("value_dtype"), FindWithExistsOp);                                     \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableInsert").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")        \
                              .TypeConstraint<V>("value_dtype"), InsertOp<true>);                                             \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableAccum").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")         \
                              .TypeConstraint<V>("value_dtype"), AccumOp<true>);                                              \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableClear").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")         \
                              .TypeConstraint<V>("value_dtype"), ClearOp);                                              \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableSize").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")          \
                              .TypeConstraint<V>("value_dtype"), SizeOp);                                               \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableExport").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")        \
                              .TypeConstraint<V>("value_dtype"), ExportOp<true, false>);                                \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableExportWithScores").Device(DEVICE_GPU)                                   \
                              .TypeConstraint<int64_t>("key_dtype").TypeConstraint<V>("value_dtype"), ExportOp<true, true>); \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableExportKeysAndScores").Device(DEVICE_GPU)                                \
                              .TypeConstraint<int64_t>("key_dtype").TypeConstraint<V>("value_dtype"), ExportOp<false, true>); \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableImport").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")        \
                              .TypeConstraint<V>("value_dtype"), ImportOp);                                             \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableSaveToFileSystem").Device(DEVICE_GPU)                                   \
                              .TypeConstraint<int64_t>("key_dtype").TypeConstraint<V>("value_dtype")                    \
                              .HostMemory("dirpath").HostMemory("file_name"), SaveToFileSystemOp);                      \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableLoadFromFileSystem").Device(DEVICE_GPU)                                 \
                              .TypeConstraint<int64_t>("key_dtype").TypeConstraint<V>("value_dtype")                    \
                              .HostMemory("dirpath").HostMemory("file_name"), LoadFromFileSystemOp);

TFRA_REGISTER_HKV(float);
TFRA_REGISTER_HKV(int8_t);
TFRA_REGISTER_HKV(int32_t);
TFRA_REGISTER_HKV(int64_t);
TFRA_REGISTER_HKV(Eigen::half);
TFRA_REGISTER_HKV(bfloat16);

#undef TFRA_REGISTER_HKV

}  // namespace tfra_mi355x
}  // namespace tensorflow
