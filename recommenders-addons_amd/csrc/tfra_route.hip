// The multi-GPU routed step, issued from C: the id-only half of the alltoall route of a batch runs ahead on a second
// stream, the step itself is local find -> alltoall(rows) -> gather and per-key sums -> alltoall(grads) -> fused
// update at the owner.  Reference: PY/shadow_embedding_ops.py:397-447 (__relocate_dense_feature__ /
// __alltoall_embedding_lookup__ route at lookup time, synchronously, through Horovod).
//
// Why C: the same sequence driven from Python (dynamic_embedding/distributed.py: RoutedPrefetchStep) is bound by the
// host — four torch.distributed calls and ~18 ctypes calls per step cost 290 us against ~110 us of device work.  Here a
// step is three C calls, and the collectives are grouped ncclSend/ncclRecv pairs on the caller's streams.
//
// Order of collectives: every collective of both channels is issued by the calling thread, in an order that depends on
// the call sequence alone — identical on every rank, so two communicators sharing a hardware queue cannot deadlock.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen()ed, nothing links against it

#include <cstring>
#include <string>
#include <vector>

#include "tfra_host.h"

using tfra::set_error;
using tfra::Table;

namespace {

// ------------------------------------------------------------------------------------------- RCCL transport
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

int load_rccl(const char* path, RcclApi* api) {
  if (!path || !*path) path = "librccl.so";
  api->lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!api->lib) return set_error(TFRA_ERR_INVALID, std::string("rccl transport: dlopen failed: ") + dlerror());
#define TFRA_SYM(field, name)                                                                         \
  api->field = reinterpret_cast<decltype(api->field)>(dlsym(api->lib, name));                         \
  if (!api->field) return set_error(TFRA_ERR_INVALID, std::string("rccl transport: missing symbol ") + name)
  TFRA_SYM(get_unique_id, "ncclGetUniqueId");
  TFRA_SYM(comm_init_rank, "ncclCommInitRank");
  TFRA_SYM(comm_destroy, "ncclCommDestroy");
  TFRA_SYM(group_start, "ncclGroupStart");
  TFRA_SYM(group_end, "ncclGroupEnd");
  TFRA_SYM(send, "ncclSend");
  TFRA_SYM(recv, "ncclRecv");
  TFRA_SYM(error_string, "ncclGetErrorString");
#undef TFRA_SYM
  return TFRA_OK;
}

struct RcclCtx {
  RcclApi api;
  ncclComm_t comm[2] = {nullptr, nullptr};
  int rank = 0, world = 1, device = 0;
};

int rccl_fail(const RcclApi& api, ncclResult_t r, const char* what) {
  return set_error(TFRA_ERR_HIP, std::string("rccl transport: ") + what + ": " + (api.error_string ? api.error_string(r) : "?"));
}

int rccl_alltoallv(void* vctx, int channel, const void* send, const size_t* send_bytes, void* recv, const size_t* recv_bytes,
                   tfra_stream_t stream) {
  RcclCtx* c = static_cast<RcclCtx*>(vctx);
  if (!c || channel < 0 || channel > 1) return set_error(TFRA_ERR_INVALID, "rccl transport: bad channel");
  hipStream_t s = (hipStream_t)stream;
  const char* sp = static_cast<const char*>(send);
  char* rp = static_cast<char*>(recv);
  ncclResult_t r = c->api.group_start();
  if (r != ncclSuccess) return rccl_fail(c->api, r, "ncclGroupStart");
  for (int peer = 0; peer < c->world; ++peer) {   // zero-byte pairs are skipped on both sides (send[a->b] == recv[b<-a])
    if (send_bytes[peer]) {
      r = c->api.send(sp, send_bytes[peer], ncclInt8, peer, c->comm[channel], s);
      if (r != ncclSuccess) { (void)c->api.group_end(); return rccl_fail(c->api, r, "ncclSend"); }
    }
    if (recv_bytes[peer]) {
      r = c->api.recv(rp, recv_bytes[peer], ncclInt8, peer, c->comm[channel], s);
      if (r != ncclSuccess) { (void)c->api.group_end(); return rccl_fail(c->api, r, "ncclRecv"); }
    }
    sp += send_bytes[peer];
    rp += recv_bytes[peer];
  }
  r = c->api.group_end();
  if (r != ncclSuccess) return rccl_fail(c->api, r, "ncclGroupEnd");
  return TFRA_OK;
}

constexpr int NSLOTS = 4;

struct Slot {
  int state = 0;                 // 0 free, 1 fed (counts on their way), 2 routed (ready event recorded)
  const int64_t* ids = nullptr;  // the caller's batch (kept alive by the caller until apply)
  size_t n = 0, u = 0, nr = 0;
  int64_t* owner_major = nullptr; // [max_n] distinct ids (the keys of plan_local) grouped by owner
  int* perm = nullptr;            // [max_n] owner-major j -> index of the key in plan_local
  int* pos2row = nullptr;         // [max_n] position -> row of the owner-major block
  int64_t* d_counts = nullptr;    // [2*world]: per-owner send counts, per-source receive counts
  int64_t* h_counts = nullptr;    // pinned copy of the 2*world counts
  int64_t* remote_ids = nullptr;  // [rcap] ids this rank serves, source-major
  size_t rcap = 0;
  tfra_sparse_plan_t* plan_local = nullptr;
  tfra_sparse_plan_t* plan_remote = nullptr;
  hipEvent_t counts_ev = nullptr, ready = nullptr, done = nullptr, ids_ev = nullptr;
  bool done_recorded = false;
  std::vector<size_t> send, recv;   // ids per peer
};

}  // namespace

struct tfra_route {
  Table* t = nullptr;
  tfra_table_t* tp = nullptr;
  bool has_tr = false;
  tfra_transport tr{};
  int world = 1, rank = 0, mode = 0, dim = 0, device = 0;
  size_t max_n = 0, row_bytes = 0;
  hipStream_t side = nullptr;
  tfra_workspace_t* ws = nullptr;
  Slot slots[NSLOTS];
  int head = 0, tail = 0, fed = 0;
  // critical-path buffers (main stream only): rows found for the other ranks, rows returned, gradient sums, gradients received
  float* rows_served = nullptr; float* grads_served = nullptr; size_t served_cap = 0;
  float* rows_back = nullptr; float* gsum = nullptr;
  std::vector<size_t> sb, rb;   // byte counts scratch
};

namespace {

int hip_fail(const char* what) { return set_error(TFRA_ERR_HIP, std::string("route: ") + what); }

template <typename T>
int dmalloc(T** p, size_t count) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T));
  if (e != hipSuccess) { *p = nullptr; return set_error(e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP, "route: hipMalloc failed"); }
  return TFRA_OK;
}

int a2a(tfra_route* r, int channel, const void* send, const std::vector<size_t>& sc, void* recv, const std::vector<size_t>& rc,
        size_t elem, hipStream_t s) {
  if (!r->has_tr) {   // one rank: what it sends is what it receives
    if (sc[0] && hipMemcpyAsync(recv, send, sc[0] * elem, hipMemcpyDeviceToDevice, s) != hipSuccess) return hip_fail("local copy");
    return TFRA_OK;
  }
  for (int i = 0; i < r->world; ++i) { r->sb[i] = sc[i] * elem; r->rb[i] = rc[i] * elem; }
  return r->tr.alltoallv(r->tr.ctx, channel, send, r->sb.data(), recv, r->rb.data(), (tfra_stream_t)s);
}

int ensure_served(tfra_route* r, size_t nr) {
  if (nr <= r->served_cap) return TFRA_OK;
  if (hipDeviceSynchronize() != hipSuccess) return hip_fail("synchronize before growing");
  (void)hipFree(r->rows_served); (void)hipFree(r->grads_served);
  r->rows_served = r->grads_served = nullptr; r->served_cap = 0;
  const size_t cap = nr + nr / 4 + 1024;
  int rc = dmalloc(&r->rows_served, cap * r->dim);
  if (!rc) rc = dmalloc(&r->grads_served, cap * r->dim);
  if (rc) return rc;
  r->served_cap = cap;
  return TFRA_OK;
}

// second half of the id-only route of a slot, once its split sizes are on the host
int finish(tfra_route* r, Slot& sl) {
  if (sl.state != 1) return TFRA_OK;
  if (hipEventSynchronize(sl.counts_ev) != hipSuccess) return hip_fail("waiting for the split sizes");
  size_t u = 0, nr = 0;
  for (int i = 0; i < r->world; ++i) {
    sl.send[i] = (size_t)sl.h_counts[i]; sl.recv[i] = (size_t)sl.h_counts[r->world + i];
    u += sl.send[i]; nr += sl.recv[i];
  }
  if (u > sl.n) return set_error(TFRA_ERR_INVALID, "route: split sizes exceed the batch (ranks out of step?)");
  if (nr > ((size_t)1 << 18)) return set_error(TFRA_ERR_UNSUPPORTED, "route: a rank serves at most 2^18 ids per batch");
  sl.u = u; sl.nr = nr;
  if (nr > sl.rcap) {
    if (hipStreamSynchronize(r->side) != hipSuccess) return hip_fail("synchronize before growing");
    (void)hipFree(sl.remote_ids); sl.remote_ids = nullptr; sl.rcap = 0;
    int rc = dmalloc(&sl.remote_ids, nr + nr / 4 + 1024);
    if (rc) return rc;
    sl.rcap = nr + nr / 4 + 1024;
  }
  int rc = ensure_served(r, nr);
  if (rc) return rc;
  rc = a2a(r, 1, sl.owner_major, sl.send, sl.remote_ids, sl.recv, sizeof(int64_t), r->side);
  if (rc) return rc;
  rc = tfra_plan_positions_to(sl.plan_local, sl.perm, sl.pos2row, (tfra_stream_t)r->side);
  if (rc) return rc;
  if (nr) {
    rc = tfra_sparse_plan_build(sl.plan_remote, nr, sl.remote_ids, r->dim, (tfra_stream_t)r->side);
    if (rc) return rc;
  }
  if (hipEventRecord(sl.ready, r->side) != hipSuccess) return hip_fail("event record");
  sl.state = 2;
  return TFRA_OK;
}

}  // namespace

extern "C" {

int tfra_rccl_unique_id(const char* librccl_path, void* id_out) {
  if (!id_out) return set_error(TFRA_ERR_INVALID, "rccl_unique_id: null out");
  RcclApi api;
  int rc = load_rccl(librccl_path, &api);
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == TFRA_RCCL_ID_BYTES, "unique id size");
  ncclUniqueId id;
  ncclResult_t r = api.get_unique_id(&id);
  if (r != ncclSuccess) return rccl_fail(api, r, "ncclGetUniqueId");
  std::memcpy(id_out, &id, sizeof(id));
  return TFRA_OK;
}

int tfra_rccl_transport_create(const char* librccl_path, const void* ids, int rank, int world, int device, tfra_transport* out) {
  if (!ids || !out || world < 1 || rank < 0 || rank >= world) return set_error(TFRA_ERR_INVALID, "rccl_transport_create: bad argument");
  RcclCtx* c = new RcclCtx();
  int rc = load_rccl(librccl_path, &c->api);
  if (rc) { delete c; return rc; }
  c->rank = rank; c->world = world; c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return hip_fail("hipSetDevice"); }
  for (int ch = 0; ch < 2; ++ch) {
    ncclUniqueId id;
    std::memcpy(&id, static_cast<const char*>(ids) + (size_t)ch * TFRA_RCCL_ID_BYTES, sizeof(id));
    ncclResult_t r = c->api.comm_init_rank(&c->comm[ch], world, id, rank);
    if (r != ncclSuccess) {
      rc = rccl_fail(c->api, r, "ncclCommInitRank");
      for (int k = 0; k < ch; ++k) (void)c->api.comm_destroy(c->comm[k]);
      delete c;
      return rc;
    }
  }
  out->ctx = c; out->rank = rank; out->world = world; out->alltoallv = rccl_alltoallv;
  return TFRA_OK;
}

int tfra_rccl_transport_destroy(tfra_transport* tr) {
  if (!tr || !tr->ctx) return TFRA_OK;
  RcclCtx* c = static_cast<RcclCtx*>(tr->ctx);
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (int ch = 0; ch < 2; ++ch) if (c->comm[ch]) (void)c->api.comm_destroy(c->comm[ch]);
  delete c;
  tr->ctx = nullptr;
  return TFRA_OK;
}

int tfra_route_destroy(tfra_route_t* r) {
  if (!r) return TFRA_OK;
  (void)hipSetDevice(r->device);
  (void)hipDeviceSynchronize();
  for (Slot& sl : r->slots) {
    (void)hipFree(sl.owner_major); (void)hipFree(sl.perm); (void)hipFree(sl.pos2row); (void)hipFree(sl.d_counts); (void)hipFree(sl.remote_ids);
    if (sl.h_counts) (void)hipHostFree(sl.h_counts);
    if (sl.plan_local) (void)tfra_sparse_plan_destroy(sl.plan_local);
    if (sl.plan_remote) (void)tfra_sparse_plan_destroy(sl.plan_remote);
    for (hipEvent_t e : {sl.counts_ev, sl.ready, sl.done, sl.ids_ev}) if (e) (void)hipEventDestroy(e);
  }
  (void)hipFree(r->rows_served); (void)hipFree(r->grads_served); (void)hipFree(r->rows_back); (void)hipFree(r->gsum);
  if (r->ws) (void)tfra_workspace_destroy(r->ws);
  if (r->side) (void)hipStreamDestroy(r->side);
  delete r;
  return TFRA_OK;
}

int tfra_route_create(tfra_table_t* table, const tfra_transport* transport, int partition_mode, size_t max_batch, tfra_route_t** out) {
  Table* t = reinterpret_cast<Table*>(table);
  if (!t || !out || max_batch == 0) return set_error(TFRA_ERR_INVALID, "route_create: bad argument");
  if (t->opts.value_dtype != TFRA_F32 || t->opts.dim % 4 != 0 || t->opts.dim > 256)
    return set_error(TFRA_ERR_UNSUPPORTED, "route_create: needs float32 rows, dim % 4 == 0, dim <= 256");
  if (max_batch > ((size_t)1 << 18)) return set_error(TFRA_ERR_UNSUPPORTED, "route_create: at most 2^18 ids per batch");
  if (transport && (!transport->alltoallv || transport->world < 1 || transport->rank < 0 || transport->rank >= transport->world))
    return set_error(TFRA_ERR_INVALID, "route_create: bad transport");
  tfra_route* r = new tfra_route();
  r->t = t; r->tp = table;
  r->has_tr = transport != nullptr;
  if (transport) { r->tr = *transport; r->world = transport->world; r->rank = transport->rank; }
  r->mode = partition_mode; r->dim = t->opts.dim; r->row_bytes = (size_t)t->opts.dim * 4; r->max_n = max_batch;
  r->device = t->opts.device;
  if (r->device < 0 && hipGetDevice(&r->device) != hipSuccess) { delete r; return hip_fail("no device"); }
  r->sb.resize(r->world); r->rb.resize(r->world);
  int rc = hipSetDevice(r->device) == hipSuccess ? TFRA_OK : hip_fail("hipSetDevice");
  if (!rc && hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking) != hipSuccess) rc = hip_fail("stream create");
  if (!rc) rc = tfra_workspace_create(r->device, &r->ws);
  const size_t n = max_batch;
  for (Slot& sl : r->slots) {
    if (rc) break;
    sl.send.assign(r->world, 0); sl.recv.assign(r->world, 0);
    rc = dmalloc(&sl.owner_major, n);
    if (!rc) rc = dmalloc(&sl.perm, n);
    if (!rc) rc = dmalloc(&sl.pos2row, n);
    if (!rc) rc = dmalloc(&sl.d_counts, (size_t)2 * r->world);
    if (!rc) rc = dmalloc(&sl.remote_ids, n);
    if (!rc) sl.rcap = n;
    if (!rc && hipHostMalloc(reinterpret_cast<void**>(&sl.h_counts), (size_t)2 * r->world * sizeof(int64_t), hipHostMallocDefault) != hipSuccess)
      rc = hip_fail("pinned allocation");
    if (!rc) rc = tfra_sparse_plan_create(r->device, &sl.plan_local);
    if (!rc) rc = tfra_sparse_plan_create(r->device, &sl.plan_remote);
    for (hipEvent_t* e : {&sl.counts_ev, &sl.ready, &sl.done, &sl.ids_ev})
      if (!rc && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = hip_fail("event create");
  }
  if (!rc) rc = dmalloc(&r->rows_back, n * r->dim);
  if (!rc) rc = dmalloc(&r->gsum, n * r->dim);
  if (!rc) rc = ensure_served(r, n);
  if (rc) { std::string keep = tfra::g_last_error; (void)tfra_route_destroy(r); tfra::g_last_error = keep; return rc; }
  *out = r;
  return TFRA_OK;
}

int tfra_route_feed(tfra_route_t* r, size_t n, const int64_t* d_ids, int ids_ready, tfra_stream_t stream) {
  if (!r) return set_error(TFRA_ERR_INVALID, "route_feed: null route");
  if (r->fed >= NSLOTS - 1) return set_error(TFRA_ERR_INVALID, "route_feed: three batches are fed ahead already");
  if (n == 0 || n > r->max_n || !d_ids) return set_error(TFRA_ERR_INVALID, "route_feed: 1 <= n <= max_batch ids expected");
  { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != r->device) { if (hipSetDevice(r->device) != hipSuccess) return hip_fail("hipSetDevice"); } }
  Slot& sl = r->slots[r->tail];
  hipStream_t side = r->side;
  if (!ids_ready) {   // the ids are still being produced on the caller's stream
    if (hipEventRecord(sl.ids_ev, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(side, sl.ids_ev, 0) != hipSuccess)
      return hip_fail("event");
  }
  if (sl.done_recorded && hipStreamWaitEvent(side, sl.done, 0) != hipSuccess) return hip_fail("event wait");   // the slot's last user
  sl.ids = d_ids; sl.n = n;
  // the de-duplication plan of the batch first: its distinct keys are what the route sends (no separate tf.unique)
  int rc = tfra_sparse_plan_build(sl.plan_local, n, d_ids, r->dim, (tfra_stream_t)side);
  if (!rc) rc = tfra_plan_partition(sl.plan_local, r->ws, r->world, r->mode, sl.owner_major, sl.perm, sl.d_counts, (tfra_stream_t)side);
  if (rc) return rc;
  if (r->has_tr) {
    for (int i = 0; i < r->world; ++i) r->sb[i] = r->rb[i] = sizeof(int64_t);
    rc = r->tr.alltoallv(r->tr.ctx, 1, sl.d_counts, r->sb.data(), sl.d_counts + r->world, r->rb.data(), (tfra_stream_t)side);
    if (rc) return rc;
  } else if (hipMemcpyAsync(sl.d_counts + 1, sl.d_counts, sizeof(int64_t), hipMemcpyDeviceToDevice, side) != hipSuccess) {
    return hip_fail("local copy");
  }
  if (hipMemcpyAsync(sl.h_counts, sl.d_counts, (size_t)2 * r->world * sizeof(int64_t), hipMemcpyDeviceToHost, side) != hipSuccess ||
      hipEventRecord(sl.counts_ev, side) != hipSuccess)
    return hip_fail("split sizes copy");
  sl.state = 1;
  r->tail = (r->tail + 1) % NSLOTS;
  r->fed += 1;
  return TFRA_OK;
}

int tfra_route_served_ids(tfra_route_t* r, const int64_t** d_ids, size_t* n, size_t* n_distinct_local) {
  if (!r || r->fed == 0) return set_error(TFRA_ERR_INVALID, "route_served_ids: no batch fed");
  Slot& sl = r->slots[r->head];
  int rc = finish(r, sl);
  if (rc) return rc;
  if (d_ids) *d_ids = sl.remote_ids;
  if (n) *n = sl.nr;
  if (n_distinct_local) *n_distinct_local = sl.u;
  return TFRA_OK;
}

int tfra_route_lookup(tfra_route_t* r, float* d_rows_out, const float* default_row, tfra_stream_t stream) {
  if (!r || r->fed == 0) return set_error(TFRA_ERR_INVALID, "route_lookup: no batch fed");
  if (!d_rows_out) return set_error(TFRA_ERR_INVALID, "route_lookup: null output");
  Slot& sl = r->slots[r->head];
  int rc = finish(r, sl);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (hipStreamWaitEvent(s, sl.ready, 0) != hipSuccess) return hip_fail("event wait");
  if (sl.nr) {
    rc = tfra_table_find(r->tp, sl.nr, sl.remote_ids, r->rows_served, nullptr, default_row, 0, stream);
    if (rc) return rc;
  }
  rc = a2a(r, 0, r->rows_served, sl.recv, r->rows_back, sl.send, r->row_bytes, s);
  if (rc) return rc;
  return tfra_gather_rows(sl.n, r->row_bytes, r->rows_back, sl.pos2row, d_rows_out, stream);
}

int tfra_route_apply(tfra_route_t* r, const tfra_opt_params* p, const float* d_grads, const float* param_default_row, tfra_stream_t stream) {
  if (!r || r->fed == 0) return set_error(TFRA_ERR_INVALID, "route_apply: no batch fed");
  if (!p || !d_grads || !param_default_row) return set_error(TFRA_ERR_INVALID, "route_apply: null argument");
  Slot& sl = r->slots[r->head];
  int rc = finish(r, sl);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (hipStreamWaitEvent(s, sl.ready, 0) != hipSuccess) return hip_fail("event wait");
  rc = tfra_plan_reduce_to(sl.plan_local, d_grads, sl.pos2row, r->gsum, stream);
  if (rc) return rc;
  rc = a2a(r, 0, r->gsum, sl.send, r->grads_served, sl.recv, r->row_bytes, s);
  if (rc) return rc;
  if (sl.nr) {
    rc = tfra_table_apply_planned(r->tp, p, sl.plan_remote, r->grads_served, param_default_row, stream);
    if (rc) return rc;
  }
  if (hipEventRecord(sl.done, s) != hipSuccess) return hip_fail("event record");
  sl.done_recorded = true;
  sl.state = 0;
  r->head = (r->head + 1) % NSLOTS;
  r->fed -= 1;
  if (r->fed) return finish(r, r->slots[r->head]);   // the next batch: its split sizes arrived during this step
  return TFRA_OK;
}

}  // extern "C"
