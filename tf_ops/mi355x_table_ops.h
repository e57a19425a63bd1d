/* The table resource and the op kernels shared by the two TensorFlow shims of this directory:
 *   hkv_ops_rocm.cc    — `TFRA>HkvHashTable*`   (ops + DEVICE_GPU kernels; replaces TFRA's _hkv_ops.so)
 *   cuckoo_ops_rocm.cc — `TFRA>CuckooHashTable*` DEVICE_GPU kernels (replaces cuckoo_hashtable_op_gpu.cu.cc, whose ops
 *                        stay registered by the reference's device-agnostic core/ops/cuckoo_hashtable_ops.cc)
 * One runtime-typed LookupInterface over the C ABI (include/tfra_mi355x.h); the flavour (bounded Hkv with scores /
 * growing cuckoo) is chosen by the creating kernel.
 */
#ifndef TFRA_MI355X_TABLE_OPS_H_
#define TFRA_MI355X_TABLE_OPS_H_
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "tensorflow/core/framework/lookup_interface.h"
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/resource_mgr.h"
#include "tensorflow/core/framework/shape_inference.h"
#include "tensorflow/core/lib/io/path.h"
#include "tensorflow/core/util/env_var.h"
#include "tfra_mi355x.h"

namespace tensorflow {
namespace tfra_mi355x {

using GPUDevice = Eigen::GpuDevice;

// =========================================== the table resource ===========================================

static int TfraDtypeOf(DataType dt) {
  switch (dt) {
    case DT_FLOAT: return TFRA_F32;
    case DT_HALF: return TFRA_F16;
    case DT_BFLOAT16: return TFRA_BF16;
    case DT_INT8: return TFRA_I8;
    case DT_INT32: return TFRA_I32;
    case DT_INT64: return TFRA_I64;
    default: return -1;
  }
}

static Status ToStatus(int rc) {
  if (rc == TFRA_OK) return OkStatus();
  const char* msg = tfra_last_error();
  if (rc == TFRA_ERR_INVALID) return errors::InvalidArgument(msg);
  if (rc == TFRA_ERR_OOM) return errors::ResourceExhausted(msg);
  return errors::Internal(msg);   // the reference maps every engine exception to kInternal (lookup_table_op_hkv.h:54-60)
}

static tfra_stream_t StreamOf(OpKernelContext* ctx) {
  return reinterpret_cast<tfra_stream_t>(ctx->eigen_device<GPUDevice>().stream());   // hipStream_t on ROCm TensorFlow
}

// device bytes come from TensorFlow's GPU allocator, like TFOrDefaultAllocator (lookup_table_op_hkv.h:329-426), so the
// table counts against `allow_growth` / the process' memory fraction instead of going around it
struct TfAllocator {
  Allocator* device;
  static void* Alloc(void* user, int kind, size_t bytes, tfra_stream_t) {
    if (kind != 0) return nullptr;   // pinned / host staging: the library's own hipHostMalloc / malloc
    return static_cast<TfAllocator*>(user)->device->AllocateRaw(256, bytes);
  }
  static void Free(void* user, int kind, void* p, tfra_stream_t) {
    if (kind == 0 && p) static_cast<TfAllocator*>(user)->device->DeallocateRaw(p);
  }
};

// One class for every value dtype: the engine takes the dtype at run time (no template per (K, V) pair).
class MI355XHashTable final : public lookup::LookupInterface {
 public:
  // cuckoo: the growing, never-evicting flavour behind `TFRA>CuckooHashTableOfTensors` (attr init_size, duplicate keys in
  // one insert allowed, last wins); else the bounded Hkv flavour with scores (HKV's unique-keys contract)
  MI355XHashTable(OpKernelContext* ctx, OpKernel* kernel, bool cuckoo) : cuckoo_(cuckoo) {
    const NodeDef& def = kernel->def();
    OP_REQUIRES_OK(ctx, GetNodeAttr(def, "key_dtype", &key_dtype_));
    OP_REQUIRES_OK(ctx, GetNodeAttr(def, "value_dtype", &value_dtype_));
    OP_REQUIRES_OK(ctx, GetNodeAttr(def, "value_shape", &value_shape_));
    OP_REQUIRES(ctx, TensorShapeUtils::IsVector(value_shape_),
                errors::InvalidArgument("Default value must be a vector, got shape ", value_shape_.DebugString()));
    // int32 keys (the reference's (int32, float) cuckoo kernels, cuckoo_hashtable_op_gpu.cu.cc:1058): widened on the device in front of
    // every engine call, narrowed behind an export, 4-byte keys in the key files (Keys64 / Export / TFRA_OPTION_KEY_BYTES_ON_DISK)
    OP_REQUIRES(ctx, (key_dtype_ == DT_INT64 || key_dtype_ == DT_INT32) && TfraDtypeOf(value_dtype_) >= 0,
                errors::InvalidArgument("hash table on MI355X: int64 / int32 keys and float / half / bfloat16 / int8 / int32 / int64 values"));
    int64_t init_capacity = 0, max_capacity = 0, max_hbm = 0, step_per_epoch = 0;
    int strategy = -1, reserved_bit = 0;
    if (cuckoo) {   // cuckoo_hashtable_op_gpu.cu.cc:58-75: init_size, 0 -> TF_HASHTABLE_INIT_SIZE, default 8192
      OP_REQUIRES_OK(ctx, GetNodeAttr(def, "init_size", &init_capacity));
      if (init_capacity == 0) {
        int64_t env_var = 0;
        ReadInt64FromEnvVar("TF_HASHTABLE_INIT_SIZE", 1024 * 8, &env_var).IgnoreError();
        init_capacity = env_var;
      }
    } else {
      OP_REQUIRES_OK(ctx, GetNodeAttr(def, "init_capacity", &init_capacity));
      OP_REQUIRES_OK(ctx, GetNodeAttr(def, "max_capacity", &max_capacity));
      OP_REQUIRES_OK(ctx, GetNodeAttr(def, "max_hbm_for_vectors", &max_hbm));
      OP_REQUIRES_OK(ctx, GetNodeAttr(def, "strategy", &strategy));
      OP_REQUIRES_OK(ctx, GetNodeAttr(def, "step_per_epoch", &step_per_epoch));
      OP_REQUIRES_OK(ctx, GetNodeAttr(def, "reserved_key_start_bit", &reserved_bit));
      OP_REQUIRES(ctx, max_hbm >= 0, errors::InvalidArgument("params max_hbm_for_vectors less than 0"));
      if (max_capacity == 0) {   // hkv_hashtable_op_gpu.cu.cc:108-119
        const char* env = std::getenv("TFRA_GPU_HASHTABLE_UPLIMIT_SIZE");
        OP_REQUIRES(ctx, env != nullptr,
                    errors::InvalidArgument("max_capaicty=0 and TFRA_GPU_HASHTABLE_UPLIMIT_SIZE not set is not valid."));
        max_capacity = std::atoll(env);
      }
    }
    tfra_table_opts o = {};
    o.struct_size = sizeof(o);
    o.value_dtype = TfraDtypeOf(value_dtype_);
    o.dim = static_cast<int32_t>(value_shape_.dim_size(0));
    o.init_capacity = static_cast<uint64_t>(init_capacity);   // 0 -> 1 Mi, max < init -> max = init: done by the engine
    o.max_capacity = static_cast<uint64_t>(max_capacity);
    o.max_hbm_for_vectors = static_cast<uint64_t>(max_hbm);
    o.strategy = strategy;
    o.step_per_epoch = step_per_epoch;
    o.reserved_key_start_bit = reserved_bit;
    o.device = -1;
    // `TFRA>HkvHashTableOfTensorsWithSlots` (fused_ops_rocm.cc): S optimizer slot vectors co-located with every row ([p|m|v]); the
    // reference's creators do not have the attrs: a plain table
    int64_t slots = 0;
    if (GetNodeAttr(def, "optimizer_slots", &slots).ok() && slots > 0) {
      OP_REQUIRES(ctx, slots <= 4, errors::InvalidArgument("optimizer_slots: at most 4"));
      o.aux_fields = static_cast<int32_t>(slots);
      std::vector<float> init;
      if (GetNodeAttr(def, "slot_init", &init).ok())
        for (size_t i = 0; i < init.size() && i < 4; ++i) o.aux_init[i] = init[i];
    }
    AllocatorAttributes attr;
    alloc_user_.device = ctx->device()->GetAllocator(attr);
    tfra_allocator bridge = {&TfAllocator::Alloc, &TfAllocator::Free, &alloc_user_};
    // TFRA_TABLE_ALLOCATOR=hip: the table's bytes come from the HIP driver instead of TensorFlow's allocator.  That is what lets a
    // table of 4 GiB or more live in a reserved address range and GROW IN PLACE (hipMemAddressReserve / hipMemMap: DESIGN §3; with a
    // caller's allocator the library can only grow by copying, which stops at a third of the HBM) — the way to reach a 10^9-slot
    // table by growth under TensorFlow.  The process must then leave the memory to the driver: TF_FORCE_GPU_ALLOW_GROWTH=true or a
    // per_process_gpu_memory_fraction below the table's share.  Default: TensorFlow's allocator, like TFOrDefaultAllocator.
    const char* from = std::getenv("TFRA_TABLE_ALLOCATOR");
    const bool hip_bytes = from != nullptr && std::string(from) == "hip";
    OP_REQUIRES_OK(ctx, ToStatus(tfra_table_create(&o, hip_bytes ? nullptr : &bridge, &table_)));
    if (key_dtype_ == DT_INT32) OP_REQUIRES_OK(ctx, ToStatus(tfra_table_set_option(table_, TFRA_OPTION_KEY_BYTES_ON_DISK, 4)));
  }
  ~MI355XHashTable() override {
    if (step_.driver) tfra_step_driver_destroy(step_.driver);
    if (ws_) tfra_workspace_destroy(ws_);
    tfra_table_destroy(table_);
  }

  size_t dim() const { return static_cast<size_t>(value_shape_.dim_size(0)); }
  tfra_table_t* raw() const { return table_; }   // for the fused ops (fused_ops_rocm.cc)
  // scratch of the front-end kernels (tfra_unique_unordered ...): one per table, ops of one table are stream-ordered
  Status Workspace(tfra_workspace_t** out) {
    mutex_lock l(aux_mu_);
    if (!ws_) TF_RETURN_IF_ERROR(ToStatus(tfra_workspace_create(-1, &ws_)));
    *out = ws_;
    return OkStatus();
  }
  // the overlapped step (tfra_table_step_overlap): its driver, and the tensors of the batches in flight — held so that their
  // buffers outlive the launches that read them
  struct StepState {
    mutex mu;
    tfra_step_driver_t* driver = nullptr;
    bool pending = false, has_next = false, has_next2 = false;
    int64_t pending_n = 0;
    Tensor cur_ids, keep_prev_ids, keep_prev_values, next, next2;
  };
  Status Step(StepState** out) {
    mutex_lock l(aux_mu_);
    if (!step_.driver) TF_RETURN_IF_ERROR(ToStatus(tfra_step_driver_create(table_, &step_.driver)));
    *out = &step_;
    return OkStatus();
  }

  // ---- LookupInterface -------------------------------------------------------------------------------
  size_t size() const override {
    size_t n = 0;
    tfra_table_size(table_, &n, nullptr);   // host result: synchronises the null stream, like the reference's private stream
    return n;
  }
  Status Find(OpKernelContext* ctx, const Tensor& keys, Tensor* values, const Tensor& default_value) override {
    return FindImpl(ctx, keys, values, default_value, nullptr);
  }
  Status FindWithExists(OpKernelContext* ctx, const Tensor& keys, Tensor* values, const Tensor& default_value, Tensor* exists) {
    return FindImpl(ctx, keys, values, default_value, exists);
  }
  Status Insert(OpKernelContext* ctx, const Tensor& keys, const Tensor& values) override {
    return InsertWithScores(ctx, keys, values, nullptr);
  }
  // `scores` = the op's int64 input, nullptr / empty tensor = none (hkv_hashtable_op_gpu.cu.cc:758-762)
  Status InsertWithScores(OpKernelContext* ctx, const Tensor& keys, const Tensor& values, const Tensor* scores) {
    const size_t n = static_cast<size_t>(keys.NumElements());
    Tensor wide;
    const int64_t* k = nullptr;
    TF_RETURN_IF_ERROR(Keys64(ctx, keys, &wide, &k));
    return ToStatus(tfra_table_insert_or_assign(table_, n, k, values.tensor_data().data(), ScoresOf(scores),
                                                UniqueFlag(), StreamOf(ctx)));
  }
  Status Accum(OpKernelContext* ctx, const Tensor& keys, const Tensor& values_or_deltas, const Tensor& exists, const Tensor* scores) {
    const size_t n = static_cast<size_t>(keys.NumElements());
    Tensor wide;
    const int64_t* k = nullptr;
    TF_RETURN_IF_ERROR(Keys64(ctx, keys, &wide, &k));
    return ToStatus(tfra_table_accum_or_assign(table_, n, k, values_or_deltas.tensor_data().data(),
                                               reinterpret_cast<const uint8_t*>(exists.tensor_data().data()), ScoresOf(scores),
                                               UniqueFlag(), StreamOf(ctx)));
  }
  Status Remove(OpKernelContext* ctx, const Tensor& keys) override {
    Tensor wide;
    const int64_t* k = nullptr;
    TF_RETURN_IF_ERROR(Keys64(ctx, keys, &wide, &k));
    return ToStatus(tfra_table_erase(table_, static_cast<size_t>(keys.NumElements()), k, StreamOf(ctx)));
  }
  Status Clear(OpKernelContext* ctx) { return ToStatus(tfra_table_clear(table_, StreamOf(ctx))); }
  Status SizeToDevice(OpKernelContext* ctx, int64_t* d_out) { return ToStatus(tfra_table_size_to_device(table_, d_out, StreamOf(ctx))); }
  Status ImportValues(OpKernelContext* ctx, const Tensor& keys, const Tensor& values) override {   // = clear + insert (:407-409)
    TF_RETURN_IF_ERROR(Clear(ctx));
    return InsertWithScores(ctx, keys, values, nullptr);
  }
  Status ExportValues(OpKernelContext* ctx) override { return Export(ctx, /*values=*/true, /*scores=*/false); }
  // keys [size], values [size, dim] (when asked), scores [size] (when asked): outputs are sized from a size read, the
  // table is then scanned once over its whole slot range (export_batch appends at a device counter)
  Status Export(OpKernelContext* ctx, bool with_values, bool with_scores) {
    size_t n = 0, capacity = 0;
    tfra_stream_t stream = StreamOf(ctx);
    TF_RETURN_IF_ERROR(ToStatus(tfra_table_size(table_, &n, stream)));
    TF_RETURN_IF_ERROR(ToStatus(tfra_table_capacity(table_, &capacity)));
    Tensor *keys = nullptr, *values = nullptr, *scores = nullptr;
    const int64_t size = static_cast<int64_t>(n);
    TF_RETURN_IF_ERROR(ctx->allocate_output("keys", TensorShape({size}), &keys));
    if (with_values) TF_RETURN_IF_ERROR(ctx->allocate_output("values", TensorShape({size, static_cast<int64_t>(dim())}), &values));
    if (with_scores) TF_RETURN_IF_ERROR(ctx->allocate_output("scores", TensorShape({size}), &scores));
    if (n == 0) return OkStatus();
    Tensor counter;
    TF_RETURN_IF_ERROR(ctx->allocate_temp(DT_UINT64, TensorShape({}), &counter));
    size_t* d_counter = reinterpret_cast<size_t*>(const_cast<char*>(counter.tensor_data().data()));
    if (hipMemsetAsync(d_counter, 0, sizeof(size_t), static_cast<hipStream_t>(stream)) != hipSuccess)
      return errors::Internal("export: hipMemsetAsync failed");
    Tensor wide;   // int32 keys: the engine exports int64 keys, narrowed into the op's output behind it
    int64_t* k64 = MutableData<int64_t>(keys);
    if (key_dtype_ == DT_INT32) {
      TF_RETURN_IF_ERROR(ctx->allocate_temp(DT_INT64, TensorShape({size}), &wide));
      k64 = MutableData<int64_t>(&wide);
    }
    TF_RETURN_IF_ERROR(ToStatus(tfra_table_export_batch(table_, capacity, 0, d_counter, k64,
                                                        values ? const_cast<char*>(values->tensor_data().data()) : nullptr,
                                                        scores ? MutableData<uint64_t>(scores) : nullptr, stream)));
    if (key_dtype_ == DT_INT32) return ToStatus(tfra_keys_narrow_i32(n, k64, MutableData<int32_t>(keys), nullptr, stream));
    return OkStatus();
  }
  Status SaveToFile(OpKernelContext* ctx, const std::string& prefix, size_t buffer_keys, bool append) {
    size_t saved = 0;
    return ToStatus(tfra_table_save(table_, prefix.c_str(), buffer_keys, append ? 1 : 0, StreamOf(ctx), &saved));
  }
  // the GPU op clears first (hkv_hashtable_op_gpu.cu.cc:619), then loads one file pair or every `<name>_mht_*` pair
  Status LoadFromFiles(OpKernelContext* ctx, const std::vector<std::string>& prefixes, size_t buffer_keys) {
    TF_RETURN_IF_ERROR(Clear(ctx));
    for (const std::string& p : prefixes) {
      size_t loaded = 0;
      TF_RETURN_IF_ERROR(ToStatus(tfra_table_load(table_, p.c_str(), buffer_keys, StreamOf(ctx), &loaded)));
    }
    return OkStatus();
  }

  DataType key_dtype() const override { return key_dtype_; }
  DataType value_dtype() const override { return value_dtype_; }
  TensorShape key_shape() const override { return TensorShape(); }
  TensorShape value_shape() const override { return value_shape_; }
  int64_t MemoryUsed() const override {
    size_t capacity = 0;
    tfra_table_capacity(table_, &capacity);
    return static_cast<int64_t>(sizeof(*this) + capacity * (16 + dim() * DataTypeSize(value_dtype_)));
  }

 private:
  template <class T>
  static const T* Data(const Tensor& t) { return reinterpret_cast<const T*>(t.tensor_data().data()); }
  template <class T>
  static T* MutableData(Tensor* t) { return reinterpret_cast<T*>(const_cast<char*>(t->tensor_data().data())); }
  // the engine's int64 keys of an op's key tensor: the tensor itself, or — int32 keys — a widened temporary (freed in stream order)
  Status Keys64(OpKernelContext* ctx, const Tensor& keys, Tensor* wide, const int64_t** out) {
    if (key_dtype_ == DT_INT64) { *out = Data<int64_t>(keys); return OkStatus(); }
    TF_RETURN_IF_ERROR(ctx->allocate_temp(DT_INT64, keys.shape(), wide));
    *out = MutableData<int64_t>(wide);
    return ToStatus(tfra_keys_widen_i32(static_cast<size_t>(keys.NumElements()), Data<int32_t>(keys), MutableData<int64_t>(wide), StreamOf(ctx)));
  }
  static const uint64_t* ScoresOf(const Tensor* scores) {
    return (scores && scores->NumElements() > 0) ? Data<uint64_t>(*scores) : nullptr;
  }
  Status FindImpl(OpKernelContext* ctx, const Tensor& keys, Tensor* values, const Tensor& default_value, Tensor* exists) {
    const size_t n = static_cast<size_t>(keys.NumElements());
    if (n == 0) return OkStatus();
    // is_full_default = (value.size() == default.size()) (hkv_hashtable_op_gpu.cu.cc:186-190)
    const int full = values->NumElements() == default_value.NumElements() ? 1 : 0;
    Tensor wide;
    const int64_t* k = nullptr;
    TF_RETURN_IF_ERROR(Keys64(ctx, keys, &wide, &k));
    return ToStatus(tfra_table_find(table_, n, k, const_cast<char*>(values->tensor_data().data()),
                                    exists ? MutableData<uint8_t>(exists) : nullptr, default_value.tensor_data().data(), full,
                                    StreamOf(ctx)));
  }

  // HKV's unique-keys contract (a bounded table evicts: one writer per key); the cuckoo ops take repeated keys, last wins
  uint32_t UniqueFlag() const { return cuckoo_ ? 0u : TFRA_FLAG_UNIQUE_KEYS; }

  bool cuckoo_ = false;
  DataType key_dtype_ = DT_INT64, value_dtype_ = DT_FLOAT;
  TensorShape value_shape_;
  TfAllocator alloc_user_{nullptr};
  tfra_table_t* table_ = nullptr;
  mutex aux_mu_;
  tfra_workspace_t* ws_ = nullptr;
  StepState step_;
};

// =========================================== op kernels ===========================================

static Status TableOf(OpKernelContext* ctx, MI355XHashTable** out, core::RefCountPtr<lookup::LookupInterface>* hold) {
  lookup::LookupInterface* table = nullptr;
  TF_RETURN_IF_ERROR(LookupResource(ctx, HandleFromInput(ctx, 0), &table));
  hold->reset(table);
  *out = dynamic_cast<MI355XHashTable*>(table);
  if (*out == nullptr) return errors::InvalidArgument("table_handle is not a TFRA MI355X hash table");
  return OkStatus();
}

// creator: one resource per (container, shared_name), deleted with the kernel when private to it
// (the reference's HashTableGpuOp, cuckoo_hashtable_op_gpu.h:43-139)
template <bool CUCKOO>
class TableOfTensorsOp : public OpKernel {
 public:
  explicit TableOfTensorsOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("use_node_name_sharing", &use_node_name_sharing_));
  }
  ~TableOfTensorsOp() override {
    if (created_ && cinfo_.resource_is_private_to_kernel())
      cinfo_.resource_manager()->Delete<lookup::LookupInterface>(cinfo_.container(), cinfo_.name()).IgnoreError();
  }
  void Compute(OpKernelContext* ctx) override {
    mutex_lock l(mu_);
    if (!created_) OP_REQUIRES_OK(ctx, cinfo_.Init(ctx->resource_manager(), def(), use_node_name_sharing_));
    lookup::LookupInterface* table = nullptr;
    OP_REQUIRES_OK(ctx, cinfo_.resource_manager()->LookupOrCreate<lookup::LookupInterface>(
                            cinfo_.container(), cinfo_.name(), &table, [ctx, this](lookup::LookupInterface** ret) {
                              lookup::LookupInterface* t = new MI355XHashTable(ctx, this, CUCKOO);
                              if (!ctx->status().ok()) {
                                t->Unref();
                                return ctx->status();
                              }
                              *ret = t;
                              return OkStatus();
                            }));
    core::ScopedUnref unref(table);
    created_ = true;
    Tensor* handle = nullptr;
    AllocatorAttributes host;
    host.set_on_host(true);
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({}), &handle, host));
    handle->scalar<ResourceHandle>()() = MakeResourceHandle<lookup::LookupInterface>(ctx, cinfo_.container(), cinfo_.name());
  }

 private:
  mutex mu_;
  bool created_ = false;
  bool use_node_name_sharing_ = false;
  ContainerInfo cinfo_;
};

#define TFRA_TABLE_OR_RETURN(ctx, t)                        \
  MI355XHashTable* t = nullptr;                             \
  core::RefCountPtr<lookup::LookupInterface> t##_hold;      \
  OP_REQUIRES_OK(ctx, TableOf(ctx, &t, &t##_hold))

static TensorShape ValuesShapeFor(const Tensor& keys, MI355XHashTable* t) {
  TensorShape s = keys.shape();
  s.AppendShape(t->value_shape());
  return s;
}

class FindOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype(), t->value_dtype()}, {t->value_dtype()}));
    const Tensor& keys = ctx->input(1);
    Tensor* values = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("values", ValuesShapeFor(keys, t), &values));
    OP_REQUIRES_OK(ctx, t->Find(ctx, keys, values, ctx->input(2)));
  }
};

class FindWithExistsOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype(), t->value_dtype()}, {t->value_dtype(), DT_BOOL}));
    const Tensor& keys = ctx->input(1);
    Tensor *values = nullptr, *exists = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("values", ValuesShapeFor(keys, t), &values));
    OP_REQUIRES_OK(ctx, ctx->allocate_output("exists", keys.shape(), &exists));
    OP_REQUIRES_OK(ctx, t->FindWithExists(ctx, keys, values, ctx->input(2), exists));
  }
};

// SCORES: the Hkv ops carry a trailing `scores: int64` input, the cuckoo ops do not
template <bool SCORES>
class InsertOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    if (SCORES) OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype(), t->value_dtype(), DT_INT64}, {}));
    else OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype(), t->value_dtype()}, {}));
    OP_REQUIRES_OK(ctx, t->CheckKeyAndValueTensorsForInsert(ctx->input(1), ctx->input(2)));
    OP_REQUIRES_OK(ctx, t->InsertWithScores(ctx, ctx->input(1), ctx->input(2), SCORES ? &ctx->input(3) : nullptr));
  }
};

template <bool SCORES>
class AccumOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    if (SCORES) OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype(), t->value_dtype(), DT_BOOL, DT_INT64}, {}));
    else OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype(), t->value_dtype(), DT_BOOL}, {}));
    OP_REQUIRES_OK(ctx, t->CheckKeyAndValueTensorsForInsert(ctx->input(1), ctx->input(2)));
    OP_REQUIRES_OK(ctx, t->Accum(ctx, ctx->input(1), ctx->input(2), ctx->input(3), SCORES ? &ctx->input(4) : nullptr));
  }
};

class RemoveOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype()}, {}));
    OP_REQUIRES_OK(ctx, t->CheckKeyTensorForRemove(ctx->input(1)));
    OP_REQUIRES_OK(ctx, t->Remove(ctx, ctx->input(1)));
  }
};

class ClearOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    OP_REQUIRES_OK(ctx, t->Clear(ctx));
  }
};

class SizeOp : public OpKernel {   // device scalar, no host sync (the reference's size_i64, :172-179)
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("size", TensorShape({}), &out));
    OP_REQUIRES_OK(ctx, t->SizeToDevice(ctx, reinterpret_cast<int64_t*>(const_cast<char*>(out->tensor_data().data()))));
  }
};

template <bool VALUES, bool SCORES>
class ExportOp : public OpKernel {   // Export / ExportWithScores / ExportKeysAndScores (split_size: one scan here)
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    OP_REQUIRES_OK(ctx, t->Export(ctx, VALUES, SCORES));
  }
};

class ImportOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype(), t->value_dtype()}, {}));
    OP_REQUIRES_OK(ctx, t->CheckKeyAndValueTensorsForImport(ctx->input(1), ctx->input(2)));
    OP_REQUIRES_OK(ctx, t->ImportValues(ctx, ctx->input(1), ctx->input(2)));
  }
};

// directory: the environment variable named by `dirpath_env` wins over the `dirpath` input (:929-941)
static Status ResolvePath(OpKernelContext* ctx, const std::string& dirpath_env, std::string* dir, std::string* file) {
  TF_RETURN_IF_ERROR(ReadStringFromEnvVar(dirpath_env, "NotFound", dir));
  if (*dir == "NotFound") {
    const Tensor& d = ctx->input(1);
    if (!TensorShapeUtils::IsScalar(d.shape())) return errors::InvalidArgument("directory path must be scalar.");
    *dir = std::string(d.scalar<tstring>()().data());
  }
  const Tensor& f = ctx->input(2);
  if (!TensorShapeUtils::IsScalar(f.shape())) return errors::InvalidArgument("file name must be scalar.");
  *file = std::string(f.scalar<tstring>()().data());
  return OkStatus();
}

class SaveToFileSystemOp : public OpKernel {
 public:
  explicit SaveToFileSystemOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("dirpath_env", &dirpath_env_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("append_to_file", &append_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("buffer_size", &buffer_size_));
  }
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    std::string dir, file;
    OP_REQUIRES_OK(ctx, ResolvePath(ctx, dirpath_env_, &dir, &file));
    OP_REQUIRES_OK(ctx, ctx->env()->RecursivelyCreateDir(dir));
    OP_REQUIRES_OK(ctx, t->SaveToFile(ctx, io::JoinPath(dir, file), static_cast<size_t>(buffer_size_), append_));
  }

 private:
  std::string dirpath_env_;
  bool append_ = false;
  int64_t buffer_size_ = 1;
};

class LoadFromFileSystemOp : public OpKernel {
 public:
  explicit LoadFromFileSystemOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("dirpath_env", &dirpath_env_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("load_entire_dir", &load_entire_dir_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("buffer_size", &buffer_size_));
  }
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    std::string dir, file;
    OP_REQUIRES_OK(ctx, ResolvePath(ctx, dirpath_env_, &dir, &file));
    std::vector<std::string> prefixes;
    if (load_entire_dir_) {   // every `<var>_mht_*` pair of the directory, each once (:1003-1019)
      const size_t sep = file.rfind("_mht_");
      std::vector<std::string> matches;
      OP_REQUIRES_OK(ctx, ctx->env()->GetMatchingPaths(io::JoinPath(dir, file.substr(0, sep + 5)) + "*", &matches));
      for (std::string& m : matches) m = m.substr(0, m.rfind('-'));   // drop the -keys / -values suffix
      std::sort(matches.begin(), matches.end());
      matches.erase(std::unique(matches.begin(), matches.end()), matches.end());
      prefixes = matches;
    } else {
      prefixes.push_back(io::JoinPath(dir, file));
    }
    OP_REQUIRES_OK(ctx, t->LoadFromFiles(ctx, prefixes, static_cast<size_t>(buffer_size_)));
  }

 private:
  std::string dirpath_env_;
  bool load_entire_dir_ = false;
  int64_t buffer_size_ = 1;
};

}  // namespace tfra_mi355x
}  // namespace tensorflow
#endif  // TFRA_MI355X_TABLE_OPS_H_
