// The multi-GPU routed step, issued from C: the id-only half of the alltoall route of a batch runs ahead on a second
// stream, the step itself is local find -> alltoall(rows) -> gather and per-key sums -> alltoall(grads) -> fused
// update at the owner.  Reference: PY/shadow_embedding_ops.py:397-447 (__relocate_dense_feature__ /
// __alltoall_embedding_lookup__ route at lookup time, synchronously, through Horovod).
//
// Why C: the same sequence driven from Python (dynamic_embedding/distributed.py: RoutedPrefetchStep) is bound by the
// host — four torch.distributed calls and ~18 ctypes calls per step cost 290 us against ~110 us of device work.  Here a
// step is three C calls, and the collectives are grouped ncclSend/ncclRecv pairs on the caller's streams.
//
// Order of collectives: every collective of both channels is issued by the CALLING thread, at points that depend on the
// call sequence alone — identical on every rank, so two communicators sharing a hardware queue cannot deadlock.  The
// helper thread only launches kernels (the plans, the partition, the position map): 15 of the ~27 launches of a step.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen()ed, nothing links against it

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
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
  decltype(&ncclCommCount) comm_count = nullptr;   // optional (reporting only)
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
  api->comm_count = reinterpret_cast<decltype(api->comm_count)>(dlsym(api->lib, "ncclCommCount"));
  return TFRA_OK;
}

struct RcclCtx {
  RcclApi api;
  ncclComm_t comm[2] = {nullptr, nullptr};
  int rank = 0, world = 1, device = 0;
};

int hip_fail(const char* what) { return set_error(TFRA_ERR_HIP, std::string("route: ") + what); }

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

int rccl_alltoallv2(void* vctx, int channel, const void* send_a, const size_t* sb_a, void* recv_a, const size_t* rb_a, const void* send_b,
                    const size_t* sb_b, void* recv_b, const size_t* rb_b, tfra_stream_t stream) {
  RcclCtx* c = static_cast<RcclCtx*>(vctx);
  if (!c || channel < 0 || channel > 1) return set_error(TFRA_ERR_INVALID, "rccl transport: bad channel");
  hipStream_t s = (hipStream_t)stream;
  ncclResult_t r = c->api.group_start();
  if (r != ncclSuccess) return rccl_fail(c->api, r, "ncclGroupStart");
  for (int set = 0; set < 2; ++set) {   // (a peer's two sends / receives are matched in the order they are issued: set a, then set b, on every rank)
    const char* sp = static_cast<const char*>(set ? send_b : send_a);
    char* rp = static_cast<char*>(set ? recv_b : recv_a);
    const size_t* sb = set ? sb_b : sb_a;
    const size_t* rb = set ? rb_b : rb_a;
    for (int peer = 0; peer < c->world; ++peer) {
      if (sb[peer]) {
        r = c->api.send(sp, sb[peer], ncclInt8, peer, c->comm[channel], s);
        if (r != ncclSuccess) { (void)c->api.group_end(); return rccl_fail(c->api, r, "ncclSend"); }
      }
      if (rb[peer]) {
        r = c->api.recv(rp, rb[peer], ncclInt8, peer, c->comm[channel], s);
        if (r != ncclSuccess) { (void)c->api.group_end(); return rccl_fail(c->api, r, "ncclRecv"); }
      }
      sp += sb[peer];
      rp += rb[peer];
    }
  }
  r = c->api.group_end();
  if (r != ncclSuccess) return rccl_fail(c->api, r, "ncclGroupEnd");
  return TFRA_OK;
}

constexpr int NSLOTS = 5;   // up to four batches fed ahead of the one being applied

// A batch moves through these stages; who issues what:
//   FED      helper : plan of the batch (CSR by key) + its distinct keys grouped by owner            [side stream]
//   COUNTED  caller : alltoall of the per-owner counts, copy to pinned memory                        [coll stream]
//   ROUTED   caller : alltoall of the ids (split sizes read on the host); then
//            helper : position -> returned-row map, plan of the ids this rank serves                 [side stream]
enum { ST_FREE = 0, ST_FED = 1, ST_COUNTED = 2, ST_ROUTED = 3 };

struct Slot {
  int state = ST_FREE;
  const int64_t* ids = nullptr;
  size_t n = 0, u = 0, nr = 0;
  int64_t* owner_major = nullptr; // [max_n] distinct ids (the keys of plan_local) grouped by owner
  int* perm = nullptr;            // [max_n] owner-major j -> index of the key in plan_local
  int* pos2row = nullptr;         // [max_n] position -> row of the owner-major block
  int64_t* d_counts = nullptr;    // [2*world]: per-owner send counts, per-source receive counts
  int64_t* h_counts = nullptr;    // pinned copy
  int64_t* remote_ids = nullptr;  // [rcap] ids this rank serves, source-major
  size_t rcap = 0;
  tfra_sparse_plan_t* plan_local = nullptr;
  tfra_sparse_plan_t* plan_remote = nullptr;
  hipEvent_t src_ev = nullptr, plan_ev = nullptr, counts_ev = nullptr, ids_ev = nullptr, ready = nullptr, done = nullptr;
  bool done_recorded = false, wait_src = false;
  std::vector<size_t> send, recv;   // ids per peer
  // helper -> caller: 1 once the job's launches are issued (and its event recorded), -1 on error
  std::atomic<int> planned{0}, posted{0};
  std::string err;
};

struct Job { int kind; int slot; };   // 0 plan, 1 post, 2 quit

// TFRA_ROUTE_TIMING=1: host time of the driver's stages, printed by tfra_route_destroy (development aid)
struct StageClock {
  static constexpr int N = 16;
  const char* name[N] = {};
  double us[N] = {};
  unsigned long calls[N] = {};
  bool on = getenv("TFRA_ROUTE_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t0;
  void start() { if (on) t0 = std::chrono::steady_clock::now(); }
  void lap(int i, const char* nm) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    name[i] = nm; us[i] += std::chrono::duration<double, std::micro>(t1 - t0).count(); calls[i] += 1; t0 = t1;
  }
  void report() {
    if (!on) return;
    for (int i = 0; i < N; ++i) if (calls[i]) fprintf(stderr, "[tfra_route] %-34s %8.2f us x %lu\n", name[i], us[i] / calls[i], calls[i]);
  }
};
StageClock g_clk;    // caller thread
StageClock g_hclk;   // helper thread

}  // namespace

struct tfra_route {
  Table* t = nullptr;
  tfra_table_t* tp = nullptr;
  bool has_tr = false, threaded = true;
  tfra_transport tr{};
  int world = 1, rank = 0, mode = 0, dim = 0, device = 0;
  size_t max_n = 0, row_bytes = 0;
  hipStream_t side = nullptr;   // kernels of the id-only half (helper thread).  (A second stream for the position map and the
                                // plan of the served ids, which are independent of the next batch's plan, made the step
                                // slower: 175 vs 130 us — more queues contending, and the split sizes arrived late.)
  hipStream_t coll = nullptr;   // its collectives (calling thread)
  tfra_workspace_t* ws = nullptr;
  Slot slots[NSLOTS];
  int head = 0, tail = 0, fed = 0;
  // critical-path buffers (main stream only): rows found for the other ranks, rows returned, gradient sums, gradients received
  float* rows_served = nullptr; float* grads_served = nullptr; size_t served_cap = 0;
  float* rows_back = nullptr; float* gsum = nullptr;
  std::vector<size_t> sb, rb;   // byte counts scratch
  // helper thread
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<Job> jobs;
};

namespace {

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

// ---- the helper's two jobs (run inline when the driver was created without a thread) --------------------------
int job_plan(tfra_route* r, Slot& sl) {
  hipStream_t side = r->side;
  g_hclk.start();
  if (sl.wait_src && hipStreamWaitEvent(side, sl.src_ev, 0) != hipSuccess) return hip_fail("event wait");
  // the slot's last user finished NSLOTS steps ago: normally complete, and then nothing is put into the stream
  if (sl.done_recorded && hipEventQuery(sl.done) != hipSuccess && hipStreamWaitEvent(side, sl.done, 0) != hipSuccess) return hip_fail("event wait");
  // the de-duplication plan of the batch first: its distinct keys are what the route sends (no separate tf.unique)
  int rc = tfra_sparse_plan_build(sl.plan_local, sl.n, sl.ids, r->dim, (tfra_stream_t)side);
  g_hclk.lap(0, "helper: plan_local build (3)");
  if (!rc) rc = tfra_plan_partition(sl.plan_local, r->ws, r->world, r->mode, sl.owner_major, sl.perm, sl.d_counts, (tfra_stream_t)side);
  g_hclk.lap(1, "helper: partition (3)");
  if (rc) return rc;
  if (hipEventRecord(sl.plan_ev, side) != hipSuccess) return hip_fail("event record");
  return TFRA_OK;
}

int job_post(tfra_route* r, Slot& sl) {
  hipStream_t side = r->side;
  g_hclk.start();
  if (hipStreamWaitEvent(side, sl.ids_ev, 0) != hipSuccess) return hip_fail("event wait");
  int rc = tfra_plan_positions_to(sl.plan_local, sl.perm, sl.pos2row, (tfra_stream_t)side);
  g_hclk.lap(2, "helper: positions_to (2)");
  if (!rc && sl.nr) rc = tfra_sparse_plan_build(sl.plan_remote, sl.nr, sl.remote_ids, r->dim, (tfra_stream_t)side);
  g_hclk.lap(3, "helper: plan_remote build (3)");
  if (rc) return rc;
  if (hipEventRecord(sl.ready, side) != hipSuccess) return hip_fail("event record");
  return TFRA_OK;
}

void run_job(tfra_route* r, const Job& j) {
  Slot& sl = r->slots[j.slot];
  const int rc = j.kind == 0 ? job_plan(r, sl) : job_post(r, sl);
  if (rc) sl.err = tfra::g_last_error;
  (j.kind == 0 ? sl.planned : sl.posted).store(rc ? -1 : 1, std::memory_order_release);
}

void worker_main(tfra_route* r) {
  (void)hipSetDevice(r->device);
  for (;;) {
    Job j;
    {
      std::unique_lock<std::mutex> lk(r->mu);
      r->cv_job.wait(lk, [&] { return !r->jobs.empty(); });
      j = r->jobs.front();
      r->jobs.pop_front();
    }
    if (j.kind == 2) return;
    run_job(r, j);
    { std::lock_guard<std::mutex> lk(r->mu); }
    r->cv_done.notify_all();
  }
}

void submit(tfra_route* r, int kind, int slot) {
  if (!r->threaded) { run_job(r, Job{kind, slot}); return; }
  { std::lock_guard<std::mutex> lk(r->mu); r->jobs.push_back(Job{kind, slot}); }
  r->cv_job.notify_one();
}

// the helper has issued the job (normally long ago: a short spin, then sleep)
int wait_issued(tfra_route* r, Slot& sl, std::atomic<int>& flag) {
  int v = flag.load(std::memory_order_acquire);
  for (int spin = 0; v == 0 && spin < 2000; ++spin) v = flag.load(std::memory_order_acquire);
  if (v == 0) {
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv_done.wait(lk, [&] { return flag.load(std::memory_order_acquire) != 0; });
    v = flag.load(std::memory_order_acquire);
  }
  if (v < 0) return set_error(TFRA_ERR_HIP, "route (helper thread): " + sl.err);
  return TFRA_OK;
}

// FED -> COUNTED: the count exchange, issued by the caller
int issue_counts(tfra_route* r, Slot& sl) {
  g_clk.start();
  int rc = wait_issued(r, sl, sl.planned);
  if (rc) return rc;
  g_clk.lap(0, "counts: wait for the helper");
  hipStream_t c = r->coll;
  if (hipStreamWaitEvent(c, sl.plan_ev, 0) != hipSuccess) return hip_fail("event wait");
  if (r->has_tr) {
    for (int i = 0; i < r->world; ++i) r->sb[i] = r->rb[i] = sizeof(int64_t);
    rc = r->tr.alltoallv(r->tr.ctx, 1, sl.d_counts, r->sb.data(), sl.d_counts + r->world, r->rb.data(), (tfra_stream_t)c);
    if (rc) return rc;
  } else if (hipMemcpyAsync(sl.d_counts + 1, sl.d_counts, sizeof(int64_t), hipMemcpyDeviceToDevice, c) != hipSuccess) {
    return hip_fail("local copy");
  }
  if (hipMemcpyAsync(sl.h_counts, sl.d_counts, (size_t)2 * r->world * sizeof(int64_t), hipMemcpyDeviceToHost, c) != hipSuccess ||
      hipEventRecord(sl.counts_ev, c) != hipSuccess)
    return hip_fail("split sizes copy");
  g_clk.lap(1, "counts: alltoall + copy + event");
  sl.state = ST_COUNTED;
  return TFRA_OK;
}

// COUNTED -> ROUTED: the id exchange once the split sizes are on the host; the rest of the id-only half goes to the helper
int issue_ids(tfra_route* r, Slot& sl) {
  g_clk.start();
  if (hipEventSynchronize(sl.counts_ev) != hipSuccess) return hip_fail("waiting for the split sizes");
  g_clk.lap(2, "ids: wait for the split sizes");
  size_t u = 0, nr = 0;
  for (int i = 0; i < r->world; ++i) {
    sl.send[i] = (size_t)sl.h_counts[i]; sl.recv[i] = (size_t)sl.h_counts[r->world + i];
    u += sl.send[i]; nr += sl.recv[i];
  }
  if (u > sl.n || u == 0) return set_error(TFRA_ERR_INVALID, "route: impossible split sizes (ranks out of step?)");
  if (nr > ((size_t)1 << 18)) return set_error(TFRA_ERR_UNSUPPORTED, "route: a rank serves at most 2^18 ids per batch");
  sl.u = u; sl.nr = nr;
  if (nr > sl.rcap) {
    if (hipDeviceSynchronize() != hipSuccess) return hip_fail("synchronize before growing");
    (void)hipFree(sl.remote_ids); sl.remote_ids = nullptr; sl.rcap = 0;
    int rc = dmalloc(&sl.remote_ids, nr + nr / 4 + 1024);
    if (rc) return rc;
    sl.rcap = nr + nr / 4 + 1024;
  }
  int rc = ensure_served(r, nr);
  if (rc) return rc;
  rc = a2a(r, 1, sl.owner_major, sl.send, sl.remote_ids, sl.recv, sizeof(int64_t), r->coll);
  if (rc) return rc;
  if (hipEventRecord(sl.ids_ev, r->coll) != hipSuccess) return hip_fail("event record");
  g_clk.lap(3, "ids: alltoall + event");
  sl.posted.store(0, std::memory_order_relaxed);
  sl.state = ST_ROUTED;
  submit(r, 1, (int)(&sl - r->slots));
  return TFRA_OK;
}

// the oldest batch must be routed before it can be looked up (stalls only when fewer than three batches are fed ahead)
int route_head(tfra_route* r) {
  Slot& sl = r->slots[r->head];
  int rc = TFRA_OK;
  if (sl.state == ST_FED) rc = issue_counts(r, sl);
  if (!rc && sl.state == ST_COUNTED) rc = issue_ids(r, sl);
  return rc;
}

// after a step: every batch fed ahead moves one stage — ids of the batches whose counts went out a step ago, then counts
// of the batches whose plan went to the helper a step ago.  Depends on the call sequence alone (same on every rank).
int advance_ahead(tfra_route* r) {
  for (int k = 0, i = r->head; k < r->fed; ++k, i = (i + 1) % NSLOTS)
    if (r->slots[i].state == ST_COUNTED) { int rc = issue_ids(r, r->slots[i]); if (rc) return rc; }
    else if (r->slots[i].state == ST_FED) { int rc = issue_counts(r, r->slots[i]); if (rc) return rc; }
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
  out->ctx = c; out->rank = rank; out->world = world; out->alltoallv = rccl_alltoallv; out->alltoallv2 = rccl_alltoallv2;
  return TFRA_OK;
}

// How many ranks the transport's communicators REALLY span, as RCCL reports it (ncclCommCount on both channels; they must agree):
// what `bench.py --gpus N` prints as rccl_ranks_seen, so that a scaling record shows that an N-rank communicator was formed.
int tfra_rccl_transport_ranks(const tfra_transport* tr, int* out) {
  if (!tr || !tr->ctx || !out) return set_error(TFRA_ERR_INVALID, "rccl transport: null argument");
  const RcclCtx* c = static_cast<const RcclCtx*>(tr->ctx);
  if (!c->api.comm_count) return set_error(TFRA_ERR_UNSUPPORTED, "rccl transport: librccl has no ncclCommCount");
  int n[2] = {0, 0};
  for (int ch = 0; ch < 2; ++ch) {
    ncclResult_t r = c->api.comm_count(c->comm[ch], &n[ch]);
    if (r != ncclSuccess) return rccl_fail(c->api, r, "ncclCommCount");
  }
  if (n[0] != n[1]) return set_error(TFRA_ERR_HIP, "rccl transport: the two channels' communicators disagree on their size");
  *out = n[0];
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
  if (r->worker.joinable()) {
    { std::lock_guard<std::mutex> lk(r->mu); r->jobs.push_back(Job{2, 0}); }
    r->cv_job.notify_one();
    r->worker.join();
  }
  g_clk.report();
  g_hclk.report();
  (void)hipSetDevice(r->device);
  (void)hipDeviceSynchronize();
  for (Slot& sl : r->slots) {
    (void)hipFree(sl.owner_major); (void)hipFree(sl.perm); (void)hipFree(sl.pos2row); (void)hipFree(sl.d_counts); (void)hipFree(sl.remote_ids);
    if (sl.h_counts) (void)hipHostFree(sl.h_counts);
    if (sl.plan_local) (void)tfra_sparse_plan_destroy(sl.plan_local);
    if (sl.plan_remote) (void)tfra_sparse_plan_destroy(sl.plan_remote);
    for (hipEvent_t e : {sl.src_ev, sl.plan_ev, sl.counts_ev, sl.ids_ev, sl.ready, sl.done}) if (e) (void)hipEventDestroy(e);
  }
  (void)hipFree(r->rows_served); (void)hipFree(r->grads_served); (void)hipFree(r->rows_back); (void)hipFree(r->gsum);
  if (r->ws) (void)tfra_workspace_destroy(r->ws);
  if (r->side) (void)hipStreamDestroy(r->side);
  if (r->coll) (void)hipStreamDestroy(r->coll);
  delete r;
  return TFRA_OK;
}

int tfra_route_create(tfra_table_t* table, const tfra_transport* transport, int partition_mode, size_t max_batch, uint32_t flags,
                      tfra_route_t** out) {
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
  r->threaded = !(flags & TFRA_ROUTE_NO_THREAD);
  if (transport) { r->tr = *transport; r->world = transport->world; r->rank = transport->rank; }
  r->mode = partition_mode; r->dim = t->opts.dim; r->row_bytes = (size_t)t->opts.dim * 4; r->max_n = max_batch;
  r->device = t->opts.device;
  if (r->device < 0 && hipGetDevice(&r->device) != hipSuccess) { delete r; return hip_fail("no device"); }
  r->sb.resize(r->world); r->rb.resize(r->world);
  int rc = hipSetDevice(r->device) == hipSuccess ? TFRA_OK : hip_fail("hipSetDevice");
  // (lowest stream priority for the id-only half changed nothing: 137 vs 131 us)
  if (!rc && (hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking) != hipSuccess ||
              hipStreamCreateWithFlags(&r->coll, hipStreamNonBlocking) != hipSuccess)) rc = hip_fail("stream create");
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
    for (hipEvent_t* e : {&sl.src_ev, &sl.plan_ev, &sl.counts_ev, &sl.ids_ev, &sl.ready, &sl.done})
      if (!rc && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = hip_fail("event create");
  }
  if (!rc) rc = dmalloc(&r->rows_back, n * r->dim);
  if (!rc) rc = dmalloc(&r->gsum, n * r->dim);
  if (!rc) rc = ensure_served(r, n);
  if (rc) { std::string keep = tfra::g_last_error; r->threaded = false; (void)tfra_route_destroy(r); tfra::g_last_error = keep; return rc; }
  if (r->threaded) r->worker = std::thread(worker_main, r);
  *out = r;
  return TFRA_OK;
}

int tfra_route_feed(tfra_route_t* r, size_t n, const int64_t* d_ids, int ids_ready, tfra_stream_t stream) {
  if (!r) return set_error(TFRA_ERR_INVALID, "route_feed: null route");
  if (r->fed >= NSLOTS - 1) return set_error(TFRA_ERR_INVALID, "route_feed: four batches are fed ahead already");
  if (n == 0 || n > r->max_n || !d_ids) return set_error(TFRA_ERR_INVALID, "route_feed: 1 <= n <= max_batch ids expected");
  { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != r->device) { if (hipSetDevice(r->device) != hipSuccess) return hip_fail("hipSetDevice"); } }
  Slot& sl = r->slots[r->tail];
  sl.wait_src = !ids_ready;   // the ids are still being produced on the caller's stream
  if (sl.wait_src && hipEventRecord(sl.src_ev, (hipStream_t)stream) != hipSuccess) return hip_fail("event record");
  sl.ids = d_ids; sl.n = n;
  sl.planned.store(0, std::memory_order_relaxed);
  sl.posted.store(0, std::memory_order_relaxed);
  sl.state = ST_FED;
  submit(r, 0, r->tail);
  r->tail = (r->tail + 1) % NSLOTS;
  r->fed += 1;
  return TFRA_OK;
}

int tfra_route_served_ids(tfra_route_t* r, const int64_t** d_ids, size_t* n, size_t* n_distinct_local) {
  if (!r || r->fed == 0) return set_error(TFRA_ERR_INVALID, "route_served_ids: no batch fed");
  Slot& sl = r->slots[r->head];
  int rc = route_head(r);
  if (rc) return rc;
  if (hipEventSynchronize(sl.ids_ev) != hipSuccess) return hip_fail("event synchronize");
  if (d_ids) *d_ids = sl.remote_ids;
  if (n) *n = sl.nr;
  if (n_distinct_local) *n_distinct_local = sl.u;
  return TFRA_OK;
}

int tfra_route_lookup(tfra_route_t* r, float* d_rows_out, const float* default_row, tfra_stream_t stream) {
  if (!r || r->fed == 0) return set_error(TFRA_ERR_INVALID, "route_lookup: no batch fed");
  if (!d_rows_out) return set_error(TFRA_ERR_INVALID, "route_lookup: null output");
  Slot& sl = r->slots[r->head];
  int rc = route_head(r);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  g_clk.start();
  if (hipStreamWaitEvent(s, sl.ids_ev, 0) != hipSuccess) return hip_fail("event wait");   // the find needs the ids only
  if (sl.nr) {
    rc = tfra_table_find(r->tp, sl.nr, sl.remote_ids, r->rows_served, nullptr, default_row, 0, stream);
    if (rc) return rc;
  }
  g_clk.lap(4, "lookup: wait + find (1)");
  rc = a2a(r, 0, r->rows_served, sl.recv, r->rows_back, sl.send, r->row_bytes, s);
  if (rc) return rc;
  g_clk.lap(5, "lookup: alltoall(rows)");
  rc = wait_issued(r, sl, sl.posted);   // the position map comes from the helper
  if (rc) return rc;
  g_clk.lap(6, "lookup: wait for the helper");
  if (hipStreamWaitEvent(s, sl.ready, 0) != hipSuccess) return hip_fail("event wait");
  rc = tfra_gather_rows(sl.n, r->row_bytes, r->rows_back, sl.pos2row, d_rows_out, stream);
  g_clk.lap(7, "lookup: wait + gather (1)");
  return rc;
}

int tfra_route_apply(tfra_route_t* r, const tfra_opt_params* p, const float* d_grads, const float* param_default_row, tfra_stream_t stream) {
  if (!r || r->fed == 0) return set_error(TFRA_ERR_INVALID, "route_apply: no batch fed");
  if (!p || !d_grads || !param_default_row) return set_error(TFRA_ERR_INVALID, "route_apply: null argument");
  Slot& sl = r->slots[r->head];
  int rc = route_head(r);
  if (!rc) rc = wait_issued(r, sl, sl.posted);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  g_clk.start();
  if (hipStreamWaitEvent(s, sl.ready, 0) != hipSuccess) return hip_fail("event wait");
  rc = tfra_plan_reduce_to(sl.plan_local, d_grads, sl.pos2row, r->gsum, stream);
  if (rc) return rc;
  g_clk.lap(8, "apply: wait + reduce_to (2)");
  rc = a2a(r, 0, r->gsum, sl.send, r->grads_served, sl.recv, r->row_bytes, s);
  if (rc) return rc;
  g_clk.lap(9, "apply: alltoall(grads)");
  if (sl.nr) {
    rc = tfra_table_apply_planned(r->tp, p, sl.plan_remote, r->grads_served, param_default_row, stream);
    if (rc) return rc;
  }
  g_clk.lap(10, "apply: apply_planned (2)");
  if (hipEventRecord(sl.done, s) != hipSuccess) return hip_fail("event record");
  g_clk.lap(11, "apply: event record");
  sl.done_recorded = true;
  sl.state = ST_FREE;
  r->head = (r->head + 1) % NSLOTS;
  r->fed -= 1;
  return advance_ahead(r);
}

}  // extern "C"
