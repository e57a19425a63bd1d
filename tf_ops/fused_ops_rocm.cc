/* TensorFlow custom ops for the FUSED paths of libtfra_mi355x.so — the part of a TFRA training step that BASELINE's north_star
 * names next to the table ops: "the sparse embedding_lookup gather + optimizer (Adam/Adagrad/FTRL) scatter-update".
 *
 * tf_ops/hkv_ops_rocm.cc gives existing TFRA graphs the reference's 14 table ops; through those alone a step still runs the
 * reference's sequence — tf.unique, (1+S) Find, gather, ResourceApply*, (1+S) Insert (PY/dynamic_embedding_optimizer.py:165-204,
 * PY/dynamic_embedding_ops.py:99-117).  The ops here put one kernel sequence of the library behind ONE op each:
 *
 *   TFRA>HkvHashTableOfTensorsWithSlots        the creator with `optimizer_slots` co-located slot vectors per row ([p|m|v]): what
 *                                              create_slots (PY/dynamic_embedding_optimizer.py:870-958) keeps in S more tables
 *   TFRA>HkvHashTableEmbeddingLookup           ids [..] -> rows [.., dim], unique ids, inverse index, count: tf.unique + Find + tf.gather
 *                                              (PY/dynamic_embedding_ops.py:99-117) with NO host read in between — every output has
 *                                              the upper-bound shape [B], the count stays on the device; the rows come from ONE find of
 *                                              all B ids (repeats hit L2: cheaper than find(U) + gather(B)), in the same kernel
 *                                              launch as the de-duplication (tfra_table_find_unique)
 *   TFRA>HkvHashTableInsertN                   Insert of the first `num` of B keys (`num` a device scalar: the lookup's count)
 *   TFRA>HkvHashTableApplySparse{Sgd,Adam,Adagrad,Ftrl}
 *                                              ids WITH repeats + their gradient rows -> duplicate sums + one fused update per key on
 *                                              the co-located rows: _resource_apply_sparse_duplicate_indices + the (1+S) finds and
 *                                              upserts around it (PY/dynamic_embedding_optimizer.py:165-204)
 *   TFRA>HkvHashTableLookupAssignStep / ...Flush
 *                                              Find(ids_i+1) and Insert(ids_i, values_i) of a streaming-assign loop as ONE kernel launch
 *                                              (csrc/tfra_step_impl.h), results those of the two ops in order
 *                                              (K/hkv_hashtable_op_gpu.cu.cc:192-213,256-267)
 *   TFRA>RcclUniqueId, TFRA>RouteCreate, TFRA>RouteFeed, TFRA>RouteLookup, TFRA>RouteApply{Sgd,Adam,Adagrad,Ftrl}
 *   TFRA>AssignRouteCreate, TFRA>AssignRouteFeed, TFRA>AssignRouteStep, TFRA>AssignRouteFlush   (round 6: lookup + insert_or_assign routed)
 *                                              HvdAllToAllEmbedding's exchange (PY/shadow_embedding_ops.py:397-447) for one table shard
 *                                              per rank, the id-only half prepared ahead, collectives = grouped ncclSend/ncclRecv
 *
 * None of them changes an existing graph; INTEGRATION.md §2.1 shows the Python that swaps them into
 * DynamicEmbeddingOptimizer._apply_op and embedding_lookup_unique.  Builds into the same _hkv_ops.so as hkv_ops_rocm.cc (one more
 * source file of that target); checked like it (tests/test_tf_shim.py: op surface pinned in tests/golden/fused_op_surface.json,
 * -fsyntax-only against tf_ops/stub/).
 */
#include <cmath>
#include <cstring>

#include "mi355x_table_ops.h"

namespace tensorflow {
namespace tfra_mi355x {

using shape_inference::InferenceContext;
using shape_inference::ShapeAndType;
using shape_inference::ShapeHandle;

// =========================================== op registrations ===========================================

static Status FusedScalarHandle(InferenceContext* c) {
  ShapeHandle h;
  return c->WithRank(c->input(0), 0, &h);
}

REGISTER_OP("TFRA>HkvHashTableOfTensorsWithSlots")
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
    .Attr("optimizer_slots: int >= 0 = 0")
    .Attr("slot_init: list(float) = []")
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

REGISTER_OP("TFRA>HkvHashTableEmbeddingLookup")
    .Input("table_handle: resource")
    .Input("ids: key_dtype")
    .Input("default_value: value_dtype")
    .Output("values: value_dtype")
    .Output("unique_ids: key_dtype")
    .Output("idx: int32")
    .Output("num_unique: int64")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(FusedScalarHandle(c));
      c->set_output(0, c->UnknownShape());
      c->set_output(1, c->UnknownShapeOfRank(1));
      c->set_output(2, c->UnknownShapeOfRank(1));
      c->set_output(3, c->Scalar());
      return OkStatus();
    });

REGISTER_OP("TFRA>HkvHashTableInsertN")
    .Input("table_handle: resource")
    .Input("keys: key_dtype")
    .Input("values: value_dtype")
    .Input("num: int64")
    .Attr("key_dtype: type")
    .Attr("value_dtype: type")
    .SetShapeFn(FusedScalarHandle);

// The hyper-parameters are scalar inputs in host memory, like TensorFlow's ResourceSparseApply* ops take them (so that a
// learning-rate schedule is a tensor, not a graph rebuild); Adam takes beta1_power / beta2_power like ResourceApplyAdam and forms
// lr_t = lr * sqrt(1 - beta2_power) / (1 - beta1_power) on the host.
REGISTER_OP("TFRA>HkvHashTableApplySparseSgd")
    .Input("table_handle: resource")
    .Input("ids: int64")
    .Input("grads: float")
    .Input("default_value: float")
    .Input("lr: float")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>HkvHashTableApplySparseAdam")
    .Input("table_handle: resource")
    .Input("ids: int64")
    .Input("grads: float")
    .Input("default_value: float")
    .Input("lr: float")
    .Input("beta1_power: float")
    .Input("beta2_power: float")
    .Input("beta1: float")
    .Input("beta2: float")
    .Input("epsilon: float")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>HkvHashTableApplySparseAdagrad")
    .Input("table_handle: resource")
    .Input("ids: int64")
    .Input("grads: float")
    .Input("default_value: float")
    .Input("lr: float")
    .Input("epsilon: float")
    .Attr("use_epsilon: bool = true")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>HkvHashTableApplySparseFtrl")
    .Input("table_handle: resource")
    .Input("ids: int64")
    .Input("grads: float")
    .Input("default_value: float")
    .Input("lr: float")
    .Input("l1: float")
    .Input("l2: float")
    .Input("lr_power: float")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>HkvHashTableLookupAssignStep")
    .Input("table_handle: resource")
    .Input("ids: int64")
    .Input("default_value: value_dtype")
    .Input("prev_values: value_dtype")
    .Input("ids_next: int64")
    .Input("ids_next2: int64")
    .Output("values: value_dtype")
    .Output("exists: bool")
    .Attr("value_dtype: type")
    .Attr("ids_were_announced: bool = false")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(FusedScalarHandle(c));
      c->set_output(0, c->UnknownShapeOfRank(2));
      c->set_output(1, c->UnknownShapeOfRank(1));
      return OkStatus();
    });

REGISTER_OP("TFRA>HkvHashTableLookupAssignFlush")
    .Input("table_handle: resource")
    .Input("prev_values: value_dtype")
    .Attr("value_dtype: type")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>RcclUniqueId")
    .Output("ids: uint8")
    .Attr("librccl_path: string = 'librccl.so'")
    .SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      c->set_output(0, c->UnknownShapeOfRank(1));
      return OkStatus();
    });

REGISTER_OP("TFRA>RouteCreate")
    .Input("table_handle: resource")
    .Input("rccl_ids: uint8")
    .Output("route_handle: resource")
    .Attr("container: string = ''")
    .Attr("shared_name: string = ''")
    .Attr("rank: int >= 0 = 0")
    .Attr("world: int >= 1 = 1")
    .Attr("partition_mode: int = 0")
    .Attr("max_batch: int >= 1 = 262144")
    .Attr("threaded: bool = true")
    .Attr("librccl_path: string = 'librccl.so'")
    .SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      c->set_output(0, c->Scalar());
      return OkStatus();
    });

REGISTER_OP("TFRA>RouteFeed")
    .Input("route_handle: resource")
    .Input("ids: int64")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>RouteLookup")
    .Input("route_handle: resource")
    .Input("default_value: float")
    .Output("values: float")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(FusedScalarHandle(c));
      c->set_output(0, c->UnknownShapeOfRank(2));
      return OkStatus();
    });

REGISTER_OP("TFRA>RouteApplySgd")
    .Input("route_handle: resource")
    .Input("grads: float")
    .Input("default_value: float")
    .Input("lr: float")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>RouteApplyAdam")
    .Input("route_handle: resource")
    .Input("grads: float")
    .Input("default_value: float")
    .Input("lr: float")
    .Input("beta1_power: float")
    .Input("beta2_power: float")
    .Input("beta1: float")
    .Input("beta2: float")
    .Input("epsilon: float")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>RouteApplyAdagrad")
    .Input("route_handle: resource")
    .Input("grads: float")
    .Input("default_value: float")
    .Input("lr: float")
    .Input("epsilon: float")
    .Attr("use_epsilon: bool = true")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>RouteApplyFtrl")
    .Input("route_handle: resource")
    .Input("grads: float")
    .Input("default_value: float")
    .Input("lr: float")
    .Input("l1: float")
    .Input("l2: float")
    .Input("lr_power: float")
    .SetShapeFn(FusedScalarHandle);

// ---- the metric's step on a hash-sharded table: lookup + insert_or_assign with ids / rows / values routed (csrc/tfra_aroute.hip) ----
// AssignRouteStep = HvdAllToAllEmbedding's lookup of the OLDEST fed batch (PY/shadow_embedding_ops.py:397-447) + the sharded
// Variable.upsert of the batch the previous Step looked up (PY/dynamic_embedding_variable.py:772-800), the owner's half as ONE launch.
REGISTER_OP("TFRA>AssignRouteCreate")
    .Input("table_handle: resource")
    .Input("rccl_ids: uint8")
    .Output("route_handle: resource")
    .Attr("container: string = ''")
    .Attr("shared_name: string = ''")
    .Attr("rank: int >= 0 = 0")
    .Attr("world: int >= 1 = 1")
    .Attr("partition_mode: int = 0")
    .Attr("max_batch: int >= 1 = 262144")
    .Attr("librccl_path: string = 'librccl.so'")
    .SetIsStateful()
    .SetShapeFn([](InferenceContext* c) {
      c->set_output(0, c->Scalar());
      return OkStatus();
    });

REGISTER_OP("TFRA>AssignRouteFeed")
    .Input("route_handle: resource")
    .Input("ids: int64")
    .SetShapeFn(FusedScalarHandle);

REGISTER_OP("TFRA>AssignRouteStep")
    .Input("route_handle: resource")
    .Input("default_value: value_dtype")
    .Input("prev_values: value_dtype")
    .Output("values: value_dtype")
    .Attr("value_dtype: type")
    .SetShapeFn([](InferenceContext* c) {
      TF_RETURN_IF_ERROR(FusedScalarHandle(c));
      c->set_output(0, c->UnknownShapeOfRank(2));
      return OkStatus();
    });

REGISTER_OP("TFRA>AssignRouteFlush")
    .Input("route_handle: resource")
    .Input("prev_values: value_dtype")
    .Attr("value_dtype: type")
    .SetShapeFn(FusedScalarHandle);

// =========================================== op kernels ===========================================

template <class T>
static const T* In(const Tensor& t) { return reinterpret_cast<const T*>(t.tensor_data().data()); }
template <class T>
static T* Out(Tensor* t) { return reinterpret_cast<T*>(const_cast<char*>(t->tensor_data().data())); }
static float HostScalar(OpKernelContext* ctx, int i) { return ctx->input(i).scalar<float>()(); }

// ---- tf.unique + Find + tf.gather as one op, no host read (PY/dynamic_embedding_ops.py:99-117) ---------------------------
// The reference de-duplicates BEFORE the lookup because a CPU (or HKV) find is paid per key; here a find of all B ids is one
// 12-us kernel whose repeats hit L2, cheaper than find(U) + gather(B): values = Find(ids) directly, and the unique ids / inverse
// index (what the backward pass and a cache-fill Insert need) come from the de-duplication of the ids running IN THE SAME LAUNCH
// (tfra_table_find_unique).  Same rows, bit for bit.
class EmbeddingLookupOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    // the outputs are allocated with the op's attr dtypes, the library writes the TABLE's row size: the two must be one
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype(), t->value_dtype()},
                                            {t->value_dtype(), t->key_dtype(), DT_INT32, DT_INT64}));
    const Tensor& ids = ctx->input(1);
    const Tensor& dflt = ctx->input(2);
    const int64_t n = ids.NumElements(), dim = static_cast<int64_t>(t->dim());
    Tensor *values = nullptr, *unique_ids = nullptr, *idx = nullptr, *num = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("values", ValuesShapeFor(ids, t), &values));
    OP_REQUIRES_OK(ctx, ctx->allocate_output("unique_ids", TensorShape({n}), &unique_ids));   // upper bound: the first num_unique are valid
    OP_REQUIRES_OK(ctx, ctx->allocate_output("idx", TensorShape({n}), &idx));
    OP_REQUIRES_OK(ctx, ctx->allocate_output("num_unique", TensorShape({}), &num));
    tfra_stream_t st = StreamOf(ctx);
    tfra_workspace_t* ws = nullptr;
    OP_REQUIRES_OK(ctx, t->Workspace(&ws));
    OP_REQUIRES(ctx, dflt.NumElements() == dim, errors::InvalidArgument("EmbeddingLookup: default_value must be one row [dim]"));
    // (no early return for an empty batch: the call zeroes num_unique, a device scalar a downstream InsertN reads)
    // ONE launch: the lookup's blocks behind the de-duplication's (find_unique_kernel, csrc/tfra_csr.hip)
    OP_REQUIRES_OK(ctx, ToStatus(tfra_table_find_unique(t->raw(), ws, static_cast<size_t>(n), In<int64_t>(ids), Out<char>(values), nullptr,
                                                        dflt.tensor_data().data(), 0, Out<int64_t>(unique_ids), Out<int32_t>(idx),
                                                        Out<int64_t>(num), st)));
  }
};

class InsertNOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->key_dtype(), t->value_dtype(), DT_INT64}, {}));
    const Tensor& keys = ctx->input(1);
    OP_REQUIRES(ctx, ctx->input(2).NumElements() == keys.NumElements() * static_cast<int64_t>(t->dim()),
                errors::InvalidArgument("InsertN: values must hold one row [dim] per key (the upper-bound shape of keys)"));
    OP_REQUIRES(ctx, ctx->input(3).NumElements() == 1, errors::InvalidArgument("InsertN: num must be a scalar"));
    OP_REQUIRES_OK(ctx, ToStatus(tfra_table_insert_or_assign_n(t->raw(), static_cast<size_t>(keys.NumElements()), In<int64_t>(ctx->input(3)),
                                                               In<int64_t>(keys), ctx->input(2).tensor_data().data(), nullptr, StreamOf(ctx))));
  }
};

// hyper-parameters of input position `first` onwards -> tfra_opt_params (host scalars)
template <int KIND>
static Status OptParamsFrom(OpKernelContext* ctx, int first, bool use_epsilon, tfra_opt_params* p) {
  std::memset(p, 0, sizeof(*p));
  p->kind = KIND;
  p->lr = HostScalar(ctx, first);
  if (KIND == TFRA_OPT_ADAM) {
    const float b1p = HostScalar(ctx, first + 1), b2p = HostScalar(ctx, first + 2);
    p->beta1 = HostScalar(ctx, first + 3); p->beta2 = HostScalar(ctx, first + 4); p->eps = HostScalar(ctx, first + 5);
    if (!(b1p < 1.0f)) return errors::InvalidArgument("ApplySparseAdam: beta1_power must be < 1 (it is beta1^t, t >= 1)");
    p->lr = p->lr * std::sqrt(1.0f - b2p) / (1.0f - b1p);   // training_ops.cc ApplyAdam: alpha
  } else if (KIND == TFRA_OPT_ADAGRAD) {
    p->eps = use_epsilon ? HostScalar(ctx, first + 1) : -1.0f;   // < 0: the TF1 rule without epsilon
  } else if (KIND == TFRA_OPT_FTRL) {
    p->l1 = HostScalar(ctx, first + 1); p->l2 = HostScalar(ctx, first + 2); p->lr_power = HostScalar(ctx, first + 3);
  }
  return OkStatus();
}

template <int KIND>
class ApplySparseOp : public OpKernel {
 public:
  explicit ApplySparseOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    if (KIND == TFRA_OPT_ADAGRAD) OP_REQUIRES_OK(ctx, ctx->GetAttr("use_epsilon", &use_epsilon_));
  }
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    const Tensor& ids = ctx->input(1);
    const Tensor& grads = ctx->input(2);
    const Tensor& dflt = ctx->input(3);
    const int64_t n = ids.NumElements(), dim = static_cast<int64_t>(t->dim());
    OP_REQUIRES(ctx, t->value_dtype() == DT_FLOAT, errors::InvalidArgument("ApplySparse*: float32 tables (half tables: reduce_by_key + apply_optimizer)"));
    OP_REQUIRES(ctx, grads.NumElements() == n * dim, errors::InvalidArgument("ApplySparse*: grads must be [ids.size, dim]"));
    OP_REQUIRES(ctx, dflt.NumElements() == dim, errors::InvalidArgument("ApplySparse*: default_value must be one row [dim]"));
    if (n == 0) return;
    tfra_opt_params p;
    OP_REQUIRES_OK(ctx, OptParamsFrom<KIND>(ctx, 4, use_epsilon_, &p));
    OP_REQUIRES_OK(ctx, ToStatus(tfra_table_apply_sparse(t->raw(), &p, static_cast<size_t>(n), In<int64_t>(ids), In<float>(grads), In<float>(dflt),
                                                         StreamOf(ctx))));
  }

 private:
  bool use_epsilon_ = true;
};

// ---- Find(ids_i+1) + Insert(ids_i, values_i) as one launch -----------------------------------------------------------------
// The table resource owns the step driver and keeps the tensors of the batches in flight alive (a TensorFlow tensor is a
// reference-counted buffer: holding the Tensor holds the bytes).  An announced batch is recognised by its ADDRESS when its own call
// comes; TensorFlow gives no such guarantee for two tensors of equal content, so with `ids_were_announced` the kernel steps the
// tensor it kept from the previous call's `ids_next` (the caller asserts that `ids` is that batch).
class LookupAssignStepOp : public OpKernel {
 public:
  explicit LookupAssignStepOp(OpKernelConstruction* ctx) : OpKernel(ctx) { OP_REQUIRES_OK(ctx, ctx->GetAttr("ids_were_announced", &announced_)); }
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    OP_REQUIRES(ctx, t->key_dtype() == DT_INT64, errors::InvalidArgument("LookupAssignStep: int64 keys"));
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, DT_INT64, t->value_dtype(), t->value_dtype(), DT_INT64, DT_INT64},
                                            {t->value_dtype(), DT_BOOL}));
    MI355XHashTable::StepState* s = nullptr;
    OP_REQUIRES_OK(ctx, t->Step(&s));
    mutex_lock l(s->mu);
    Tensor ids = ctx->input(1);
    if (announced_ && s->has_next) {
      OP_REQUIRES(ctx, s->next.NumElements() == ids.NumElements(), errors::InvalidArgument("LookupAssignStep: ids is not the batch announced as ids_next"));
      ids = s->next;
    }
    const Tensor& dflt = ctx->input(2);
    const Tensor& prev_values = ctx->input(3);
    const Tensor& nxt = ctx->input(4);
    const Tensor& nx2 = ctx->input(5);
    const int64_t n = ids.NumElements(), dim = static_cast<int64_t>(t->dim());
    OP_REQUIRES(ctx, n > 0, errors::InvalidArgument("LookupAssignStep: empty batch (LookupAssignFlush writes the pending batch back)"));
    OP_REQUIRES(ctx, dflt.NumElements() == dim, errors::InvalidArgument("LookupAssignStep: default_value must be one row [dim]"));
    OP_REQUIRES(ctx, !s->pending || prev_values.NumElements() == s->pending_n * dim,
                errors::InvalidArgument("LookupAssignStep: prev_values must hold one row per id of the previous call"));
    Tensor *values = nullptr, *exists = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("values", TensorShape({n, dim}), &values));
    OP_REQUIRES_OK(ctx, ctx->allocate_output("exists", TensorShape({n}), &exists));
    // a second announced batch that the previous call already saw as its ids_next2 is stepped at the kept address too
    Tensor next = nxt;
    if (announced_ && s->has_next2 && s->next2.NumElements() == nxt.NumElements()) next = s->next2;
    OP_REQUIRES_OK(ctx, ToStatus(tfra_table_step_overlap(
                            s->driver, static_cast<size_t>(n), In<int64_t>(ids), Out<char>(values), Out<uint8_t>(exists), dflt.tensor_data().data(), 0,
                            s->pending ? prev_values.tensor_data().data() : nullptr, nullptr, static_cast<size_t>(next.NumElements()),
                            next.NumElements() ? In<int64_t>(next) : nullptr, static_cast<size_t>(nx2.NumElements()),
                            nx2.NumElements() ? In<int64_t>(nx2) : nullptr, StreamOf(ctx))));
    // alive until the launch that reads them has been enqueued AND the next call has replaced them: ids / values of the batch
    // being written back (read by this launch), this batch (written back by the next), the two announced batches
    s->keep_prev_ids = s->cur_ids;
    s->keep_prev_values = prev_values;
    s->cur_ids = ids;
    s->pending = true;
    s->pending_n = n;
    s->next = next; s->has_next = next.NumElements() > 0;
    s->next2 = nx2; s->has_next2 = nx2.NumElements() > 0;
  }

 private:
  bool announced_ = false;
};

class LookupAssignFlushOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_TABLE_OR_RETURN(ctx, t);
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, t->value_dtype()}, {}));
    MI355XHashTable::StepState* s = nullptr;
    OP_REQUIRES_OK(ctx, t->Step(&s));
    mutex_lock l(s->mu);
    if (!s->pending) return;
    const Tensor& prev_values = ctx->input(1);
    OP_REQUIRES(ctx, prev_values.NumElements() == s->pending_n * static_cast<int64_t>(t->dim()),
                errors::InvalidArgument("LookupAssignFlush: prev_values must hold one row per id of the previous call"));
    OP_REQUIRES_OK(ctx, ToStatus(tfra_table_step_overlap_flush(s->driver, prev_values.tensor_data().data(), nullptr, StreamOf(ctx))));
    s->keep_prev_ids = s->cur_ids;
    s->keep_prev_values = prev_values;
    s->pending = false;
    s->has_next = s->has_next2 = false;
  }
};

// ---- the multi-GPU route ----------------------------------------------------------------------------------------------------
class RcclUniqueIdOp : public OpKernel {   // rank 0 runs it; the host framework broadcasts the 2 x 128 bytes (hvd.broadcast)
 public:
  explicit RcclUniqueIdOp(OpKernelConstruction* ctx) : OpKernel(ctx) { OP_REQUIRES_OK(ctx, ctx->GetAttr("librccl_path", &path_)); }
  void Compute(OpKernelContext* ctx) override {
    Tensor* out = nullptr;
    AllocatorAttributes host;
    host.set_on_host(true);
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({2 * TFRA_RCCL_ID_BYTES}), &out, host));
    for (int ch = 0; ch < 2; ++ch)
      OP_REQUIRES_OK(ctx, ToStatus(tfra_rccl_unique_id(path_.c_str(), Out<char>(out) + ch * TFRA_RCCL_ID_BYTES)));
  }

 private:
  std::string path_;
};

class RouteResource final : public ResourceBase {
 public:
  RouteResource() = default;
  ~RouteResource() override {
    if (route) tfra_route_destroy(route);
    if (has_transport) tfra_rccl_transport_destroy(&transport);
    if (table) table->Unref();
  }
  std::string DebugString() const override { return "TFRA MI355X route"; }
  tfra_route_t* route = nullptr;
  tfra_transport transport = {};
  bool has_transport = false;
  lookup::LookupInterface* table = nullptr;   // one reference held: the shard outlives its route
  int64_t dim = 0;
  mutex mu;
  std::vector<Tensor> fed;                    // ids of the batches fed and not yet applied (the driver reads them on its own streams)
};

class RouteCreateOp : public OpKernel {
 public:
  explicit RouteCreateOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("rank", &rank_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("world", &world_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("partition_mode", &mode_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("max_batch", &max_batch_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("threaded", &threaded_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("librccl_path", &path_));
  }
  void Compute(OpKernelContext* ctx) override {
    mutex_lock l(mu_);
    if (!created_) OP_REQUIRES_OK(ctx, cinfo_.Init(ctx->resource_manager(), def(), true));
    RouteResource* res = nullptr;
    OP_REQUIRES_OK(ctx, cinfo_.resource_manager()->LookupOrCreate<RouteResource>(
                            cinfo_.container(), cinfo_.name(), &res, [ctx, this](RouteResource** ret) {
                              MI355XHashTable* t = nullptr;
                              core::RefCountPtr<lookup::LookupInterface> hold;
                              TF_RETURN_IF_ERROR(TableOf(ctx, &t, &hold));
                              if (t->value_dtype() != DT_FLOAT) return errors::InvalidArgument("RouteCreate: float32 tables");
                              RouteResource* r = new RouteResource();
                              t->Ref();
                              r->table = t;
                              r->dim = static_cast<int64_t>(t->dim());
                              if (world_ > 1) {
                                const Tensor& ids = ctx->input(1);
                                if (ids.NumElements() != 2 * TFRA_RCCL_ID_BYTES) { r->Unref(); return errors::InvalidArgument("RouteCreate: rccl_ids must be TFRA>RcclUniqueId's output (256 bytes)"); }
                                Status s = ToStatus(tfra_rccl_transport_create(path_.c_str(), ids.tensor_data().data(), static_cast<int>(rank_),
                                                                               static_cast<int>(world_), -1, &r->transport));
                                if (!s.ok()) { r->Unref(); return s; }
                                r->has_transport = true;
                              }
                              Status s = ToStatus(tfra_route_create(t->raw(), r->has_transport ? &r->transport : nullptr, static_cast<int>(mode_),
                                                                    static_cast<size_t>(max_batch_), threaded_ ? 0u : TFRA_ROUTE_NO_THREAD, &r->route));
                              if (!s.ok()) { r->Unref(); return s; }
                              *ret = r;
                              return OkStatus();
                            }));
    core::ScopedUnref unref(res);
    created_ = true;
    Tensor* handle = nullptr;
    AllocatorAttributes host;
    host.set_on_host(true);
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({}), &handle, host));
    handle->scalar<ResourceHandle>()() = MakeResourceHandle<RouteResource>(ctx, cinfo_.container(), cinfo_.name());
  }

 private:
  mutex mu_;
  bool created_ = false;
  ContainerInfo cinfo_;
  int64_t rank_ = 0, world_ = 1, mode_ = 0, max_batch_ = 262144;
  bool threaded_ = true;
  std::string path_;
};

#define TFRA_ROUTE_OR_RETURN(ctx, r)                                                   \
  RouteResource* r = nullptr;                                                          \
  OP_REQUIRES_OK(ctx, LookupResource(ctx, HandleFromInput(ctx, 0), &r));               \
  core::ScopedUnref r##_unref(r)

class RouteFeedOp : public OpKernel {   // from the input pipeline's prefetch stage, up to three batches ahead
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_ROUTE_OR_RETURN(ctx, r);
    mutex_lock l(r->mu);
    const Tensor& ids = ctx->input(1);
    OP_REQUIRES_OK(ctx, ToStatus(tfra_route_feed(r->route, static_cast<size_t>(ids.NumElements()), In<int64_t>(ids), /*ids_ready=*/0, StreamOf(ctx))));
    r->fed.push_back(ids);
  }
};

class RouteLookupOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_ROUTE_OR_RETURN(ctx, r);
    mutex_lock l(r->mu);
    OP_REQUIRES(ctx, !r->fed.empty(), errors::InvalidArgument("RouteLookup: no batch has been fed"));
    const Tensor& dflt = ctx->input(1);
    OP_REQUIRES(ctx, dflt.NumElements() == r->dim, errors::InvalidArgument("RouteLookup: default_value must be one row [dim]"));
    Tensor* values = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("values", TensorShape({r->fed.front().NumElements(), r->dim}), &values));
    OP_REQUIRES_OK(ctx, ToStatus(tfra_route_lookup(r->route, Out<float>(values), In<float>(dflt), StreamOf(ctx))));
  }
};

template <int KIND>
class RouteApplyOp : public OpKernel {
 public:
  explicit RouteApplyOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    if (KIND == TFRA_OPT_ADAGRAD) OP_REQUIRES_OK(ctx, ctx->GetAttr("use_epsilon", &use_epsilon_));
  }
  void Compute(OpKernelContext* ctx) override {
    TFRA_ROUTE_OR_RETURN(ctx, r);
    mutex_lock l(r->mu);
    OP_REQUIRES(ctx, !r->fed.empty(), errors::InvalidArgument("RouteApply*: no batch is waiting for its gradients"));
    const Tensor& grads = ctx->input(1);
    const Tensor& dflt = ctx->input(2);
    OP_REQUIRES(ctx, grads.NumElements() == r->fed.front().NumElements() * r->dim, errors::InvalidArgument("RouteApply*: grads must be [batch, dim]"));
    OP_REQUIRES(ctx, dflt.NumElements() == r->dim, errors::InvalidArgument("RouteApply*: default_value must be one row [dim]"));
    tfra_opt_params p;
    OP_REQUIRES_OK(ctx, OptParamsFrom<KIND>(ctx, 3, use_epsilon_, &p));
    OP_REQUIRES_OK(ctx, ToStatus(tfra_route_apply(r->route, &p, In<float>(grads), In<float>(dflt), StreamOf(ctx))));
    r->fed.erase(r->fed.begin());   // retired: its ids may go
  }

 private:
  bool use_epsilon_ = true;
};

// ---- the routed assign step (tfra_assign_route_*) ---------------------------------------------------------------------------------
class AssignRouteResource final : public ResourceBase {
 public:
  AssignRouteResource() = default;
  ~AssignRouteResource() override {
    if (route) tfra_assign_route_destroy(route);
    if (has_transport) tfra_rccl_transport_destroy(&transport);
    if (table) table->Unref();
  }
  std::string DebugString() const override { return "TFRA MI355X assign route"; }
  tfra_assign_route_t* route = nullptr;
  tfra_transport transport = {};
  bool has_transport = false;
  MI355XHashTable* table = nullptr;           // one reference held: the shard outlives its route
  int64_t dim = 0;
  mutex mu;
  std::vector<Tensor> fed;                    // ids of the batches fed and not yet looked up (the driver reads them on its own stream)
  Tensor pending_ids, keep_ids, keep_values;  // the batch looked up last (written back by the next Step) and what the launch in flight reads
  bool pending = false;
  int64_t pending_n = 0;
};

class AssignRouteCreateOp : public OpKernel {
 public:
  explicit AssignRouteCreateOp(OpKernelConstruction* ctx) : OpKernel(ctx) {
    OP_REQUIRES_OK(ctx, ctx->GetAttr("rank", &rank_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("world", &world_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("partition_mode", &mode_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("max_batch", &max_batch_));
    OP_REQUIRES_OK(ctx, ctx->GetAttr("librccl_path", &path_));
  }
  void Compute(OpKernelContext* ctx) override {
    mutex_lock l(mu_);
    if (!created_) OP_REQUIRES_OK(ctx, cinfo_.Init(ctx->resource_manager(), def(), true));
    AssignRouteResource* res = nullptr;
    OP_REQUIRES_OK(ctx, cinfo_.resource_manager()->LookupOrCreate<AssignRouteResource>(
                            cinfo_.container(), cinfo_.name(), &res, [ctx, this](AssignRouteResource** ret) {
                              MI355XHashTable* t = nullptr;
                              core::RefCountPtr<lookup::LookupInterface> hold;
                              TF_RETURN_IF_ERROR(TableOf(ctx, &t, &hold));
                              AssignRouteResource* r = new AssignRouteResource();
                              t->Ref();
                              r->table = t;
                              r->dim = static_cast<int64_t>(t->dim());
                              if (world_ > 1) {
                                const Tensor& ids = ctx->input(1);
                                if (ids.NumElements() != 2 * TFRA_RCCL_ID_BYTES) { r->Unref(); return errors::InvalidArgument("AssignRouteCreate: rccl_ids must be TFRA>RcclUniqueId's output (256 bytes)"); }
                                Status s = ToStatus(tfra_rccl_transport_create(path_.c_str(), ids.tensor_data().data(), static_cast<int>(rank_),
                                                                               static_cast<int>(world_), -1, &r->transport));
                                if (!s.ok()) { r->Unref(); return s; }
                                r->has_transport = true;
                              }
                              Status s = ToStatus(tfra_assign_route_create(t->raw(), r->has_transport ? &r->transport : nullptr, static_cast<int>(mode_),
                                                                           static_cast<size_t>(max_batch_), &r->route));
                              if (!s.ok()) { r->Unref(); return s; }
                              *ret = r;
                              return OkStatus();
                            }));
    core::ScopedUnref unref(res);
    created_ = true;
    Tensor* handle = nullptr;
    AllocatorAttributes host;
    host.set_on_host(true);
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, TensorShape({}), &handle, host));
    handle->scalar<ResourceHandle>()() = MakeResourceHandle<AssignRouteResource>(ctx, cinfo_.container(), cinfo_.name());
  }

 private:
  mutex mu_;
  bool created_ = false;
  ContainerInfo cinfo_;
  int64_t rank_ = 0, world_ = 1, mode_ = 0, max_batch_ = 262144;
  std::string path_;
};

#define TFRA_AROUTE_OR_RETURN(ctx, r)                                                  \
  AssignRouteResource* r = nullptr;                                                    \
  OP_REQUIRES_OK(ctx, LookupResource(ctx, HandleFromInput(ctx, 0), &r));               \
  core::ScopedUnref r##_unref(r)

class AssignRouteFeedOp : public OpKernel {   // from the input pipeline's prefetch stage, up to five batches ahead of their Step
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_AROUTE_OR_RETURN(ctx, r);
    mutex_lock l(r->mu);
    const Tensor& ids = ctx->input(1);
    OP_REQUIRES_OK(ctx, ToStatus(tfra_assign_route_feed(r->route, static_cast<size_t>(ids.NumElements()), In<int64_t>(ids), /*ids_ready=*/0, StreamOf(ctx))));
    r->fed.push_back(ids);
  }
};

class AssignRouteStepOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_AROUTE_OR_RETURN(ctx, r);
    mutex_lock l(r->mu);
    OP_REQUIRES(ctx, !r->fed.empty(), errors::InvalidArgument("AssignRouteStep: no batch has been fed"));
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, r->table->value_dtype(), r->table->value_dtype()}, {r->table->value_dtype()}));
    const Tensor& dflt = ctx->input(1);
    const Tensor& prev_values = ctx->input(2);
    OP_REQUIRES(ctx, dflt.NumElements() == r->dim, errors::InvalidArgument("AssignRouteStep: default_value must be one row [dim]"));
    OP_REQUIRES(ctx, !r->pending || prev_values.NumElements() == r->pending_n * r->dim,
                errors::InvalidArgument("AssignRouteStep: prev_values must hold one row per id of the batch the previous Step looked up"));
    const Tensor ids = r->fed.front();
    Tensor* values = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output("values", TensorShape({ids.NumElements(), r->dim}), &values));
    OP_REQUIRES_OK(ctx, ToStatus(tfra_assign_route_step(r->route, Out<char>(values), dflt.tensor_data().data(),
                                                        r->pending ? prev_values.tensor_data().data() : nullptr, StreamOf(ctx))));
    // alive until the launches that read them have run: the ids written back by this call, its values; this batch's ids until the next
    r->keep_ids = r->pending_ids;
    r->keep_values = prev_values;
    r->pending_ids = ids;
    r->pending = true;
    r->pending_n = ids.NumElements();
    r->fed.erase(r->fed.begin());
  }
};

class AssignRouteFlushOp : public OpKernel {
 public:
  using OpKernel::OpKernel;
  void Compute(OpKernelContext* ctx) override {
    TFRA_AROUTE_OR_RETURN(ctx, r);
    mutex_lock l(r->mu);
    if (!r->pending) return;
    OP_REQUIRES_OK(ctx, ctx->MatchSignature({DT_RESOURCE, r->table->value_dtype()}, {}));
    const Tensor& prev_values = ctx->input(1);
    OP_REQUIRES(ctx, prev_values.NumElements() == r->pending_n * r->dim,
                errors::InvalidArgument("AssignRouteFlush: prev_values must hold one row per id of the batch the last Step looked up"));
    OP_REQUIRES_OK(ctx, ToStatus(tfra_assign_route_flush(r->route, prev_values.tensor_data().data(), StreamOf(ctx))));
    r->keep_ids = r->pending_ids;
    r->keep_values = prev_values;
    r->pending = false;
  }
};

// =========================================== kernel registrations ===========================================
// The table-typed ops for K = int64 x the reference's GPU value types (hkv_hashtable_op_gpu.cu.cc:1133-1138); the optimizer ops
// are float32 (the planned write-back), hyper-parameters and RCCL ids in host memory.

#define TFRA_REGISTER_FUSED(V)                                                                                                  \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableOfTensorsWithSlots").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")    \
                              .TypeConstraint<V>("value_dtype"), TableOfTensorsOp<false>);                                      \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableEmbeddingLookup").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")       \
                              .TypeConstraint<V>("value_dtype"), EmbeddingLookupOp);                                            \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableInsertN").Device(DEVICE_GPU).TypeConstraint<int64_t>("key_dtype")               \
                              .TypeConstraint<V>("value_dtype"), InsertNOp);                                                    \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableLookupAssignStep").Device(DEVICE_GPU).TypeConstraint<V>("value_dtype"),         \
                          LookupAssignStepOp);                                                                                  \
  REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableLookupAssignFlush").Device(DEVICE_GPU).TypeConstraint<V>("value_dtype"),        \
                          LookupAssignFlushOp);                                                                                 \
  REGISTER_KERNEL_BUILDER(Name("TFRA>AssignRouteStep").Device(DEVICE_GPU).TypeConstraint<V>("value_dtype"), AssignRouteStepOp); \
  REGISTER_KERNEL_BUILDER(Name("TFRA>AssignRouteFlush").Device(DEVICE_GPU).TypeConstraint<V>("value_dtype"), AssignRouteFlushOp);

TFRA_REGISTER_FUSED(float);
TFRA_REGISTER_FUSED(int8_t);
TFRA_REGISTER_FUSED(int32_t);
TFRA_REGISTER_FUSED(int64_t);
TFRA_REGISTER_FUSED(Eigen::half);
TFRA_REGISTER_FUSED(bfloat16);
#undef TFRA_REGISTER_FUSED

REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableApplySparseSgd").Device(DEVICE_GPU).HostMemory("lr"), ApplySparseOp<TFRA_OPT_SGD>);
REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableApplySparseAdam").Device(DEVICE_GPU).HostMemory("lr").HostMemory("beta1_power")
                            .HostMemory("beta2_power").HostMemory("beta1").HostMemory("beta2").HostMemory("epsilon"),
                        ApplySparseOp<TFRA_OPT_ADAM>);
REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableApplySparseAdagrad").Device(DEVICE_GPU).HostMemory("lr").HostMemory("epsilon"),
                        ApplySparseOp<TFRA_OPT_ADAGRAD>);
REGISTER_KERNEL_BUILDER(Name("TFRA>HkvHashTableApplySparseFtrl").Device(DEVICE_GPU).HostMemory("lr").HostMemory("l1").HostMemory("l2")
                            .HostMemory("lr_power"), ApplySparseOp<TFRA_OPT_FTRL>);

REGISTER_KERNEL_BUILDER(Name("TFRA>RcclUniqueId").Device(DEVICE_GPU).HostMemory("ids"), RcclUniqueIdOp);
REGISTER_KERNEL_BUILDER(Name("TFRA>RouteCreate").Device(DEVICE_GPU).HostMemory("rccl_ids").HostMemory("route_handle"), RouteCreateOp);
REGISTER_KERNEL_BUILDER(Name("TFRA>AssignRouteCreate").Device(DEVICE_GPU).HostMemory("rccl_ids").HostMemory("route_handle"), AssignRouteCreateOp);
REGISTER_KERNEL_BUILDER(Name("TFRA>AssignRouteFeed").Device(DEVICE_GPU), AssignRouteFeedOp);
REGISTER_KERNEL_BUILDER(Name("TFRA>RouteFeed").Device(DEVICE_GPU), RouteFeedOp);
REGISTER_KERNEL_BUILDER(Name("TFRA>RouteLookup").Device(DEVICE_GPU), RouteLookupOp);
REGISTER_KERNEL_BUILDER(Name("TFRA>RouteApplySgd").Device(DEVICE_GPU).HostMemory("lr"), RouteApplyOp<TFRA_OPT_SGD>);
REGISTER_KERNEL_BUILDER(Name("TFRA>RouteApplyAdam").Device(DEVICE_GPU).HostMemory("lr").HostMemory("beta1_power").HostMemory("beta2_power")
                            .HostMemory("beta1").HostMemory("beta2").HostMemory("epsilon"), RouteApplyOp<TFRA_OPT_ADAM>);
REGISTER_KERNEL_BUILDER(Name("TFRA>RouteApplyAdagrad").Device(DEVICE_GPU).HostMemory("lr").HostMemory("epsilon"), RouteApplyOp<TFRA_OPT_ADAGRAD>);
REGISTER_KERNEL_BUILDER(Name("TFRA>RouteApplyFtrl").Device(DEVICE_GPU).HostMemory("lr").HostMemory("l1").HostMemory("l2").HostMemory("lr_power"),
                        RouteApplyOp<TFRA_OPT_FTRL>);

}  // namespace tfra_mi355x
}  // namespace tensorflow
