// The metric's step — lookup(B ids) + insert_or_assign(B ids, B rows) — on a table hash-sharded over the GPUs of one node.
// Reference shape: HvdAllToAllEmbedding / __alltoall_embedding_lookup__ (PY/shadow_embedding_ops.py:397-447: unique ->
// partition by owner -> alltoall(ids) -> local lookup -> alltoall(rows) -> stitch) and, for the write-back, the same route taken by
// the values (`Variable.upsert` partitions keys AND values by owner: PY/dynamic_embedding_variable.py:772-800).
//
// What is different here (MI355X-first, not the reference's call pattern):
//   * the OWNER's half of a step is the overlapped step of tfra_step_impl.h — lookup of the ids it serves for batch i+1 and
//     write-back of the rows it received for batch i in ONE launch (ids the two share are forwarded from the received rows);
//   * everything that depends on the ids alone runs AHEAD of the step, off its critical path, and is TWO launches per batch:
//     the route plan (routeplan_insert_kernel + routeplan_emit_kernel) de-duplicates the batch (the distinct ids are what travels: a Zipf-1.2 batch of 131 072 ids has ~22 K),
//     groups the distinct ids by owner, records the last position of every distinct id (insert_or_assign: the last occurrence
//     wins) and the position -> returned-row map.  The route of tfra_route.hip needs 11 launches for the same (CSR plan 3,
//     partition 3, position map 2, plan of the served ids 3);
//   * the count exchange, the one host read of the split sizes and the id exchange of a batch happen one / two steps before the
//     batch is looked up (channel 1 of the transport, its own stream), so the owner knows the ids it will serve TWO batches ahead —
//     which is what the overlapped step wants (ids_next / ids_next2: its de-duplication plans are built inside the step launches).
// Left on a step's critical path, all on the caller's stream:
//     pack (values of batch i at the last positions, owner-major: one gather) -> alltoall(values) -> step launch at the owner ->
//     alltoall(rows) -> one gather (owner-major rows -> the batch's positions).
// Results: exactly those of ONE table that sees, per step, lookup(all ranks' ids) and then insert_or_assign(rank 0's batch, rank 1's
// batch, ...) — a key written by several ranks in one step keeps the highest rank's last occurrence (the owner receives the
// rows source-major and the last one wins), and lookup i+1 sees every write of step i (Insert exclusive, Find shared:
// K/hkv_hashtable_op_gpu.cu.cc:192-213,256-267).
//
// Collectives: every alltoall of both channels is issued by the CALLING thread at points that depend on the call sequence
// alone — identical on every rank.  No helper thread: the id-only half is two launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tfra_host.h"

using namespace tfra;

namespace {

int hip_fail(const char* what) { return set_error(TFRA_ERR_HIP, std::string("assign route: ") + what); }

// ------------------------------------------------------------------------------------------- the id-only half: TWO launches
// For a batch of n ids (<= 2^18):
//   keys_out[0 .. U)   the distinct ids grouped by owner (owner 0's first), order inside a group unspecified
//   lastpos[j]         the LAST position of keys_out[j] in the batch
//   pos2row[p]         the row j with keys_out[j] == ids[p]
//   d_counts[w]        distinct ids owned by rank w (int64: what the count exchange sends)
// routeplan_insert_kernel (blocks of 1024 threads, one or two ids per thread):
//   A  equal ids of a block meet in an LDS hash table (compare-and-swap on the key, max on position + 1);
//   B  every distinct id of the block goes into a global open-addressing table (compare-and-swap; max on position + 1); the block
//      whose swap installed a key counts it under its owner in LDS, draws the block's base per owner from the per-owner counters
//      (ONE returned add per block and owner: 128 x world adds per batch, not one per key) and stores (owner, base + index) in
//      the key's entry: the key's row inside its owner's group.
// routeplan_emit_kernel (256 threads per block; launched behind it: every count is final, no block waits for another):
//   position blocks: probe the table for ids[p] -> pos2row[p] = prefix[owner] + row inside the group;
//   slot blocks: one table slot per thread -> keys_out / lastpos of the key it holds; the same thread empties its slot of the OTHER
//      of the two tables (they alternate: the table of the build before is clean again when the next build needs it).
// (A first form did all of this in ONE launch with a grid-wide meeting — every block co-resident, thread 0 of each polling an arrival
// counter.  Its 2048 waves held a quarter of the chip's wave slots while they waited, beside the step launch they are meant to stay
// out of the way of, and a kernel that needs all its blocks resident at once is one more thing that can go wrong on a GPU shared
// with foreign work.  Two plain launches cost one more enqueue on a stream that is not the critical path.)
// The two sentinel key values (EMPTY_KEY, LOCKED_KEY) have slots of their own behind the table.
constexpr int RP_NT = 1024;
constexpr unsigned RP_MAX_WORLD = 64;
constexpr size_t RP_MAX_IDS = (size_t)1 << 18;
struct RpEnt { i64 key; unsigned pos1, row1; };   // key (EMPTY_KEY = free) | last position + 1 | owner << 20 | (row inside the owner's group + 1)
constexpr unsigned RP_ROW_BITS = 20, RP_ROW_MASK = (1u << RP_ROW_BITS) - 1u;

struct RpArgs {
  unsigned n, m2, world, mode;
  const i64* ids;
  RpEnt* ent;            // this build's table
  RpEnt* old_ent;        // the other one: emptied by the emit kernel
  unsigned* gcount;      // [RP_MAX_WORLD] distinct ids per owner of this build (zero before it)
  unsigned* gcount_next; // the counters of the NEXT build (two sets alternate): zeroed by the emit kernel
  i64* keys_out; int* lastpos; int* pos2row; i64* d_counts;
  i64* h_counts;         // pinned: the same counts for the host (the split sizes of the id exchange) — no copy command
  unsigned pos_blocks;   // emit: blocks that map positions (the rest scan slots)
};

__device__ __forceinline__ int rp_owner_of(i64 key, unsigned num, unsigned mode) {   // default_partition_fn, PY/dynamic_embedding_variable.py:165-197
  if (mode == 0) return (int)((unsigned)(key & 0x7fffffff) % num);                    // (as tfra_partition: CUDA-build branch / CPU-build branch / opt-in mix)
  if (mode == 1) { i64 m = key % (i64)num; return (int)(m < 0 ? m + (i64)num : m); }
  return (int)__umul64hi(fmix64((u64)key), (u64)num);
}
__device__ __forceinline__ unsigned rp_home(i64 key, unsigned m2) { return (unsigned)(fmix64((u64)key) >> 20) & (m2 - 1); }

template <int IPT>
__global__ __launch_bounds__(RP_NT) void routeplan_insert_kernel(const RpArgs a) {
  constexpr unsigned LDSN = 2048u * IPT;   // slots of the block's LDS table: two per id
  constexpr int NR = 2 * IPT;
  __shared__ i64 s_key[LDSN];
  __shared__ unsigned s_pos[LDSN + 2];
  __shared__ unsigned s_own[RP_MAX_WORLD], s_obase[RP_MAX_WORLD];
  const unsigned tid = threadIdx.x, bid = blockIdx.x, m2 = a.m2;
  for (unsigned i = tid; i < LDSN + 2; i += RP_NT) { if (i < LDSN) s_key[i] = EMPTY_KEY; s_pos[i] = 0; }
  if (tid < RP_MAX_WORLD) { s_own[tid] = 0; s_obase[tid] = 0; }
  __syncthreads();
  // ---- A ------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const unsigned g = (bid * IPT + q) * RP_NT + tid;
    if (g < a.n) {
      const i64 id = a.ids[g];
      unsigned slot;
      if (is_reserved_key(id)) slot = LDSN + (unsigned)reserved_index(id);
      else {
        slot = (unsigned)(fmix64((u64)id) >> 41) & (LDSN - 1);
        for (;;) {
          const i64 was = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&s_key[slot]), (unsigned long long)EMPTY_KEY, (unsigned long long)id);
          if (was == EMPTY_KEY || was == id) break;
          slot = (slot + 1) & (LDSN - 1);
        }
      }
      atomicMax(&s_pos[slot], g + 1u);
    }
  }
  __syncthreads();
  // ---- B: NR + 1 items per thread — the block's LDS slots tid, tid + 1024, ...; threads 0 and 1 also carry the two sentinel slots -----
  unsigned myslot[NR + 1], myidx[NR + 1], myown[NR + 1];
  bool mine[NR + 1];
#pragma unroll
  for (int r = 0; r <= NR; ++r) {
    const bool sentinel = r == NR;
    const unsigned ls = sentinel ? LDSN + (tid & 1u) : tid + (unsigned)r * RP_NT;
    const bool have = (!sentinel || tid < 2) && s_pos[ls] != 0;
    mine[r] = false; myidx[r] = 0; myown[r] = 0; myslot[r] = 0;
    if (!have) continue;
    const i64 key = sentinel ? EMPTY_KEY + (i64)(tid & 1u) : s_key[ls];
    if (sentinel) {
      myslot[r] = m2 + (tid & 1u);
      const i64 w = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&a.ent[myslot[r]].key), (unsigned long long)EMPTY_KEY, 1ULL);
      mine[r] = w == EMPTY_KEY;
    } else {
      unsigned sl = rp_home(key, m2);
      for (;;) {
        const i64 w = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&a.ent[sl].key), (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (w == EMPTY_KEY) { mine[r] = true; break; }
        if (w == key) break;
        sl = (sl + 1) & (m2 - 1);
      }
      myslot[r] = sl;
    }
    atomicMax(&a.ent[myslot[r]].pos1, s_pos[ls]);
    if (mine[r]) {
      myown[r] = (unsigned)rp_owner_of(key, a.world, a.mode);
      myidx[r] = atomicAdd(&s_own[myown[r]], 1u);
    }
  }
  __syncthreads();
  if (tid < a.world && s_own[tid]) s_obase[tid] = atomicAdd(a.gcount + tid, s_own[tid]);
  __syncthreads();
#pragma unroll
  for (int r = 0; r <= NR; ++r)
    if (mine[r]) a.ent[myslot[r]].row1 = (myown[r] << RP_ROW_BITS) | (s_obase[myown[r]] + myidx[r] + 1u);
}

__global__ __launch_bounds__(256) void routeplan_emit_kernel(const RpArgs a) {
  __shared__ unsigned s_pref[RP_MAX_WORLD];
  const unsigned tid = threadIdx.x, bid = blockIdx.x, m2 = a.m2;
  if (tid < a.world) s_pref[tid] = a.gcount[tid];
  __syncthreads();
  if (tid == 0) {
    unsigned run = 0;
    for (unsigned w = 0; w < a.world; ++w) {
      const unsigned c = s_pref[w];
      if (bid == 0) { a.d_counts[w] = (i64)c; __hip_atomic_store(a.h_counts + w, (i64)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      s_pref[w] = run;
      run += c;
    }
  }
  if (bid == 0 && tid < RP_MAX_WORLD) a.gcount_next[tid] = 0;
  __syncthreads();
  if (bid < a.pos_blocks) {
    const unsigned p = bid * 256u + tid;
    if (p >= a.n) return;
    const i64 id = a.ids[p];
    unsigned sl;
    if (is_reserved_key(id)) sl = m2 + (unsigned)reserved_index(id);
    else {
      sl = rp_home(id, m2);
      for (unsigned it = 0; it < m2 && a.ent[sl].key != id; ++it) sl = (sl + 1) & (m2 - 1);   // (present: the insert kernel put it there)
    }
    const unsigned v = a.ent[sl].row1;
    // (v == 0 or a foreign entry: the ids changed between the two launches — a caller that fed a buffer still being written without saying
    // so (ids_ready = 0).  The results are then undefined, but the row stays inside the buffers.)
    const unsigned own = min(v >> RP_ROW_BITS, a.world - 1u), loc = v & RP_ROW_MASK;
    a.pos2row[p] = loc ? (int)min(s_pref[own] + loc - 1u, a.n - 1u) : 0;
    return;
  }
  const unsigned i = (bid - a.pos_blocks) * 256u + tid;
  if (i >= m2 + 2) return;
  const uint4 e = *reinterpret_cast<const uint4*>(a.ent + i);
  const i64 k = (i64)(((u64)e.y << 32) | e.x);
  if (k != EMPTY_KEY) {
    const unsigned row = s_pref[e.w >> RP_ROW_BITS] + (e.w & RP_ROW_MASK) - 1u;
    a.keys_out[row] = i >= m2 ? EMPTY_KEY + (i64)(i - m2) : k;
    a.lastpos[row] = (int)e.z - 1;
  }
  *reinterpret_cast<uint4*>(a.old_ent + i) = make_uint4(0u, 0x80000000u, 0u, 0u);   // {EMPTY_KEY, 0, 0}
}

__global__ __launch_bounds__(256) void rp_fill_kernel(RpEnt* e, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    *reinterpret_cast<uint4*>(e + i) = make_uint4(0u, 0x80000000u, 0u, 0u);
}

template <typename T>
int dmalloc(T** p, size_t count) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T));
  if (e != hipSuccess) { *p = nullptr; return set_error(e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP, "assign route: hipMalloc failed"); }
  return TFRA_OK;
}

// ------------------------------------------------------------------------------------------- the driver
constexpr int NS = 12;  // slots of the ring.  At most MAX_LIVE batches are in flight (the one being written back, the one looked up, up to
                        // five fed ahead); the other slots are slack: a slot is reused NS - MAX_LIVE + 1 steps after its batch was written
                        // back, so the route plan of a new batch — which waits for that write-back — finds it complete although the host
                        // runs a few steps ahead of the GPU (with 8 slots the split sizes of 9 batches in 10 were waited for on the host)
constexpr int MAX_LIVE = 7;
// A batch moves through these stages (everything is issued by the calling thread, in call order; the id-only half on ONE stream of the
// driver's own, `ahead`, in order — route plan, count exchange, copy of the split sizes, id exchange need no events between them):
//   FED      the route plan's two launches                                                   [ahead]
//   COUNTED  alltoall of the per-owner counts, copy to pinned memory                         [ahead]
//   ROUTED   split sizes read on the host, alltoall of the distinct ids                      [ahead]
//   LOOKED   its lookup has been issued; the next step writes it back and retires it
enum { ST_FREE = 0, ST_FED = 1, ST_COUNTED = 2, ST_ROUTED = 3, ST_LOOKED = 4 };

struct ASlot {
  int state = ST_FREE;
  const int64_t* ids = nullptr;
  size_t n = 0, u = 0, nr = 0;
  i64* owner_major = nullptr;   // [max_n] distinct ids grouped by owner
  int* lastpos = nullptr;       // [max_n] last position of owner_major[j]
  int* pos2row = nullptr;       // [max_n] position -> owner-major row
  i64* d_counts = nullptr;      // [2 * world] per-owner send counts | per-source receive counts
  i64* h_counts = nullptr;      // pinned copy
  i64* recv_buf = nullptr;      // [rcap] the ids this rank serves for the batch, source-major (one rank without a transport: unused,
  size_t rcap = 0;              //  what is sent IS what is received: recv_ids = owner_major)
  i64* recv_ids = nullptr;
  hipEvent_t src_ev = nullptr, done = nullptr;
  hipEvent_t counts_ev = nullptr, ids_ev = nullptr;   // borrowed from the driver's ring of marks (mark_ahead)
  bool done_recorded = false, wait_src = false, main_waited = false;
  std::vector<size_t> send, recv;   // ids per peer
};

}  // namespace

struct tfra_assign_route {
  Table* t = nullptr;
  tfra_table_t* tp = nullptr;
  tfra_step_driver_t* drv = nullptr;
  bool has_tr = false;
  tfra_transport tr{};
  int world = 1, rank = 0, mode = 0, device = 0;
  size_t max_n = 0, row_bytes = 0;
  hipStream_t ahead = nullptr;
  static constexpr int NMARK = 4 * NS;   // a mark is referenced for at most ~3 steps, at most two are recorded per call
  hipEvent_t marks[NMARK] = {};
  int mark_next = 0;
  // route-plan scratch: two tables that alternate (the emit kernel of a build empties the other one), two sets of per-owner counters
  RpEnt* ent[2] = {nullptr, nullptr};
  unsigned* gcount = nullptr;     // [2][RP_MAX_WORLD]
  unsigned m2 = 0, builds = 0;
  ASlot slots[NS];
  int head = 0, look = 0, tail = 0;   // oldest live slot (looked up, not yet written back — or == look), next to look up, next free
  int live = 0;                        // slots between head and tail
  bool pending = false;                // slots[head] has been looked up and awaits its write-back
  // critical-path buffers (caller's stream only).  One rank without a transport: vrecv = vsend and rows_back = rows_served
  unsigned char* vsend = nullptr; unsigned char* rows_back = nullptr;            // [max_n rows]
  unsigned char* vrecv = nullptr; unsigned char* rows_served = nullptr; size_t served_cap = 0;   // [served_cap rows]
  std::vector<size_t> sb, rb, cb;
  unsigned long long n_steps = 0, n_stalls = 0;
};

namespace {

int a2a(tfra_assign_route* r, int channel, const void* send, const std::vector<size_t>& sc, void* recv, const std::vector<size_t>& rc, size_t elem,
        hipStream_t s) {
  if (!r->has_tr) return TFRA_OK;   // one rank: the receive buffer IS the send buffer (see create)
  for (int i = 0; i < r->world; ++i) { r->sb[i] = sc[i] * elem; r->rb[i] = rc[i] * elem; }
  return r->tr.alltoallv(r->tr.ctx, channel, send, r->sb.data(), recv, r->rb.data(), (tfra_stream_t)s);
}

int ensure_served(tfra_assign_route* r, size_t nr) {
  if (!r->has_tr || nr <= r->served_cap) return TFRA_OK;
  if (hipDeviceSynchronize() != hipSuccess) return hip_fail("synchronize before growing");
  (void)hipFree(r->vrecv); (void)hipFree(r->rows_served);
  r->vrecv = r->rows_served = nullptr; r->served_cap = 0;
  const size_t cap = std::min(RP_MAX_IDS, nr + nr / 4 + 1024);
  int rc = dmalloc(&r->vrecv, cap * r->row_bytes);
  if (!rc) rc = dmalloc(&r->rows_served, cap * r->row_bytes);
  if (rc) return rc;
  r->served_cap = cap;
  return TFRA_OK;
}

// FED: the two launches of the id-only half
int issue_plan(tfra_assign_route* r, ASlot& sl) {
  hipStream_t st = r->ahead;
  if (sl.wait_src && hipStreamWaitEvent(st, sl.src_ev, 0) != hipSuccess) return hip_fail("event wait");
  // the slot's buffers were last read by the step that wrote its previous batch back (NS - 1 steps ago: normally complete)
  if (sl.done_recorded && hipEventQuery(sl.done) != hipSuccess && hipStreamWaitEvent(st, sl.done, 0) != hipSuccess) return hip_fail("event wait");
  const unsigned p = r->builds & 1u;
  r->builds += 1;
  RpArgs a{};
  a.n = (unsigned)sl.n; a.m2 = r->m2; a.world = (unsigned)r->world; a.mode = (unsigned)r->mode;
  a.ids = (const i64*)sl.ids;
  a.ent = r->ent[p]; a.old_ent = r->ent[p ^ 1u];
  a.gcount = r->gcount + (size_t)p * RP_MAX_WORLD; a.gcount_next = r->gcount + (size_t)(p ^ 1u) * RP_MAX_WORLD;
  a.keys_out = sl.owner_major; a.lastpos = sl.lastpos; a.pos2row = sl.pos2row; a.d_counts = sl.d_counts; a.h_counts = sl.h_counts;
  a.pos_blocks = (unsigned)((sl.n + 255) / 256);
  static const int ipt_env = [] { const char* e = getenv("TFRA_RP_IPT"); return e ? atoi(e) : 0; }();
  const int ipt = (ipt_env == 1 || ipt_env == 2 || ipt_env == 4) ? ipt_env : (sl.n <= (size_t)32 * RP_NT ? 1 : 4);
  const unsigned iblk = (unsigned)((sl.n + (size_t)RP_NT * ipt - 1) / ((size_t)RP_NT * ipt));
  if (ipt == 1) routeplan_insert_kernel<1><<<iblk, RP_NT, 0, st>>>(a);
  else if (ipt == 2) routeplan_insert_kernel<2><<<iblk, RP_NT, 0, st>>>(a);
  else routeplan_insert_kernel<4><<<iblk, RP_NT, 0, st>>>(a);
  routeplan_emit_kernel<<<a.pos_blocks + (r->m2 + 2 + 255) / 256, 256, 0, st>>>(a);
  if (hipGetLastError() != hipSuccess) return hip_fail("route-plan launch failed");
  sl.state = ST_FED;
  return TFRA_OK;
}

// FED -> COUNTED.  defer_exchange: the caller issues the count exchange itself, together with another batch's id exchange (one group)
int issue_counts(tfra_assign_route* r, ASlot& sl, bool defer_exchange = false) {
  hipStream_t c = r->ahead;
  if (r->has_tr && !defer_exchange) {
    for (int i = 0; i < r->world; ++i) r->sb[i] = r->rb[i] = sizeof(int64_t);
    int rc = r->tr.alltoallv(r->tr.ctx, 1, sl.d_counts, r->sb.data(), sl.d_counts + r->world, r->rb.data(), (tfra_stream_t)c);
    if (rc) return rc;
  }
  // the send counts reach pinned memory from the route plan itself; what arrived from the other ranks is copied (one rank: the receive
  // count is the send count, filled in on the host)
  if (r->has_tr && hipMemcpyAsync(sl.h_counts + r->world, sl.d_counts + r->world, (size_t)r->world * sizeof(int64_t), hipMemcpyDeviceToHost, c) != hipSuccess)
    return hip_fail("split sizes copy");
  sl.counts_ev = nullptr;   // = the event the caller records behind this stage (mark_ahead)
  sl.state = ST_COUNTED;
  return TFRA_OK;
}

// COUNTED -> ROUTED: the ONE host read of a batch (the split sizes, as hvd.alltoall(ids, splits) needs them too), then the ids.
// with_counts_of: a FED batch whose count exchange travels in the same group as this batch's ids (tfra_transport::alltoallv2)
int issue_ids(tfra_assign_route* r, ASlot& sl, ASlot* with_counts_of = nullptr) {
  if (!sl.counts_ev) return set_error(TFRA_ERR_HIP, "assign route: internal: stage without an event");
  if (hipEventQuery(sl.counts_ev) != hipSuccess) {
    r->n_stalls += 1;
    if (hipEventSynchronize(sl.counts_ev) != hipSuccess) return hip_fail("waiting for the split sizes");
  }
  if (!r->has_tr) sl.h_counts[1] = sl.h_counts[0];
  size_t u = 0, nr = 0;
  for (int i = 0; i < r->world; ++i) {
    sl.send[i] = (size_t)sl.h_counts[i]; sl.recv[i] = (size_t)sl.h_counts[r->world + i];
    u += sl.send[i]; nr += sl.recv[i];
  }
  if (u > sl.n || u == 0) return set_error(TFRA_ERR_INVALID, "assign route: impossible split sizes (ranks out of step?)");
  if (nr > RP_MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "assign route: a rank serves at most 2^18 ids per batch");
  sl.u = u; sl.nr = nr;
  int rc = TFRA_OK;
  if (r->has_tr && nr > sl.rcap) {
    if (hipDeviceSynchronize() != hipSuccess) return hip_fail("synchronize before growing");
    (void)hipFree(sl.recv_buf); sl.recv_buf = nullptr; sl.rcap = 0;
    const size_t cap = std::min(RP_MAX_IDS, nr + nr / 4 + 1024);
    rc = dmalloc(&sl.recv_buf, cap);
    if (rc) return rc;
    sl.rcap = cap;
  }
  sl.recv_ids = r->has_tr ? sl.recv_buf : sl.owner_major;
  rc = ensure_served(r, nr);
  if (rc) return rc;
  if (with_counts_of && r->has_tr && r->tr.alltoallv2) {
    std::vector<size_t>& cb = r->cb;
    for (int i = 0; i < r->world; ++i) { r->sb[i] = sl.send[i] * sizeof(int64_t); r->rb[i] = sl.recv[i] * sizeof(int64_t); cb[i] = sizeof(int64_t); }
    rc = r->tr.alltoallv2(r->tr.ctx, 1, sl.owner_major, r->sb.data(), sl.recv_ids, r->rb.data(), with_counts_of->d_counts, cb.data(),
                          with_counts_of->d_counts + r->world, cb.data(), (tfra_stream_t)r->ahead);
  } else {
    rc = a2a(r, 1, sl.owner_major, sl.send, sl.recv_ids, sl.recv, sizeof(int64_t), r->ahead);
    if (!rc && with_counts_of && r->has_tr) {
      for (int i = 0; i < r->world; ++i) r->sb[i] = r->rb[i] = sizeof(int64_t);
      rc = r->tr.alltoallv(r->tr.ctx, 1, with_counts_of->d_counts, r->sb.data(), with_counts_of->d_counts + r->world, r->rb.data(), (tfra_stream_t)r->ahead);
    }
  }
  if (rc) return rc;
  sl.ids_ev = nullptr;   // = the event the caller records behind this stage (mark_ahead)
  sl.main_waited = false;
  sl.state = ST_ROUTED;
  return TFRA_OK;
}

// ONE event behind everything a call put into the ahead stream: it stands for the end of every stage issued since the last one
// (the stream is in order; a stage's consumer waits a little longer than it must — two steps ahead of its use, it does not matter)
int mark_ahead(tfra_assign_route* r) {
  bool any = false;
  for (ASlot& sl : r->slots) any = any || (sl.state == ST_COUNTED && !sl.counts_ev) || (sl.state == ST_ROUTED && !sl.ids_ev);
  if (!any) return TFRA_OK;
  hipEvent_t e = r->marks[r->mark_next];
  r->mark_next = (r->mark_next + 1) % tfra_assign_route::NMARK;
  if (hipEventRecord(e, r->ahead) != hipSuccess) return hip_fail("event record");
  for (ASlot& sl : r->slots) {
    if (sl.state == ST_COUNTED && !sl.counts_ev) sl.counts_ev = e;
    if (sl.state == ST_ROUTED && !sl.ids_ev) sl.ids_ev = e;
  }
  return TFRA_OK;
}

int ensure_routed(tfra_assign_route* r, ASlot& sl) {
  int rc = TFRA_OK;
  if (sl.state == ST_FED) { rc = issue_counts(r, sl); if (!rc) rc = mark_ahead(r); }
  if (!rc && sl.state == ST_COUNTED) { rc = issue_ids(r, sl); if (!rc) rc = mark_ahead(r); }
  return rc;
}

// after a step: every batch fed ahead moves one stage.  Depends on the call sequence alone (the same on every rank).
int advance_ahead(tfra_assign_route* r) {
  // the normal case of a full pipeline: ONE batch has its split sizes and gets its ids routed, the NEXT one gets its counts exchanged —
  // both exchanges as one group (one RCCL kernel instead of two)
  ASlot* to_route = nullptr;
  ASlot* to_count = nullptr;
  int n_route = 0, n_count = 0;
  for (int k = 0, i = r->head; k < r->live; ++k, i = (i + 1) % NS) {
    ASlot& sl = r->slots[i];
    if (sl.state == ST_COUNTED) { if (!to_route) to_route = &sl; n_route += 1; }
    else if (sl.state == ST_FED) { if (!to_count) to_count = &sl; n_count += 1; }
  }
  if (n_route == 1 && n_count == 1) {
    // (the received counts are copied to pinned memory BEHIND the exchange: issue_counts with the exchange deferred puts only that copy
    // into the stream, so it comes after the group below)
    int rc = issue_ids(r, *to_route, to_count);
    if (!rc) rc = issue_counts(r, *to_count, /*defer_exchange=*/true);
    if (rc) return rc;
    return mark_ahead(r);
  }
  for (int k = 0, i = r->head; k < r->live; ++k, i = (i + 1) % NS) {
    ASlot& sl = r->slots[i];
    if (sl.state == ST_COUNTED) { int rc = issue_ids(r, sl); if (rc) return rc; }
    else if (sl.state == ST_FED) { int rc = issue_counts(r, sl); if (rc) return rc; }
  }
  return mark_ahead(r);
}

// write-back half of a step: the rows of the batch in slots[head] (looked up by the previous step) travel to their owners
int send_values(tfra_assign_route* r, ASlot& pv, const void* values_prev, hipStream_t s) {
  int rc = tfra_gather_rows(pv.u, r->row_bytes, values_prev, pv.lastpos, r->vsend, (tfra_stream_t)s);   // last occurrence of every distinct id, owner-major
  if (rc) return rc;
  return a2a(r, 0, r->vsend, pv.send, r->vrecv, pv.recv, r->row_bytes, s);
}

void retire_head(tfra_assign_route* r, hipStream_t s) {
  ASlot& pv = r->slots[r->head];
  if (hipEventRecord(pv.done, s) == hipSuccess) pv.done_recorded = true;
  pv.state = ST_FREE;
  r->head = (r->head + 1) % NS;
  r->live -= 1;
  r->pending = false;
}

}  // namespace

extern "C" {

int tfra_assign_route_destroy(tfra_assign_route_t* r) {
  if (!r) return TFRA_OK;
  (void)hipSetDevice(r->device);
  (void)hipDeviceSynchronize();
  if (r->drv) (void)tfra_step_driver_destroy(r->drv);
  for (ASlot& sl : r->slots) {
    (void)hipFree(sl.owner_major); (void)hipFree(sl.lastpos); (void)hipFree(sl.pos2row); (void)hipFree(sl.d_counts); (void)hipFree(sl.recv_buf);
    if (sl.h_counts) (void)hipHostFree(sl.h_counts);
    for (hipEvent_t e : {sl.src_ev, sl.done}) if (e) (void)hipEventDestroy(e);
  }
  for (hipEvent_t e : r->marks) if (e) (void)hipEventDestroy(e);
  for (int p = 0; p < 2; ++p) (void)hipFree(r->ent[p]);
  (void)hipFree(r->gcount);
  (void)hipFree(r->vsend); (void)hipFree(r->rows_served);
  if (r->has_tr) { (void)hipFree(r->rows_back); (void)hipFree(r->vrecv); }   // (aliases of the two above otherwise)
  if (r->ahead) (void)hipStreamDestroy(r->ahead);
  delete r;
  return TFRA_OK;
}

int tfra_assign_route_create(tfra_table_t* table, const tfra_transport* transport, int partition_mode, size_t max_batch,
                             tfra_assign_route_t** out) {
  Table* t = reinterpret_cast<Table*>(table);
  if (!t || !out || max_batch == 0) return set_error(TFRA_ERR_INVALID, "assign_route_create: bad argument");
  if (max_batch > RP_MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "assign_route_create: at most 2^18 ids per batch");
  if (partition_mode < 0 || partition_mode > 2) return set_error(TFRA_ERR_INVALID, "assign_route_create: partition_mode 0, 1 or 2");
  if (transport && (!transport->alltoallv || transport->world < 1 || transport->rank < 0 || transport->rank >= transport->world))
    return set_error(TFRA_ERR_INVALID, "assign_route_create: bad transport");
  if (transport && (unsigned)transport->world > RP_MAX_WORLD) return set_error(TFRA_ERR_UNSUPPORTED, "assign_route_create: at most 64 ranks");
  tfra_assign_route* r = new tfra_assign_route();
  r->t = t; r->tp = table;
  r->has_tr = transport != nullptr;
  if (transport) { r->tr = *transport; r->world = transport->world; r->rank = transport->rank; }
  r->mode = partition_mode; r->row_bytes = t->field_bytes; r->max_n = max_batch;
  r->device = t->opts.device;
  if (r->device < 0 && hipGetDevice(&r->device) != hipSuccess) { delete r; return hip_fail("no device"); }
  r->sb.resize(r->world); r->rb.resize(r->world); r->cb.resize(r->world);
  int rc = hipSetDevice(r->device) == hipSuccess ? TFRA_OK : hip_fail("hipSetDevice");
  if (!rc && hipStreamCreateWithFlags(&r->ahead, hipStreamNonBlocking) != hipSuccess) rc = hip_fail("stream create");
  if (!rc) rc = tfra_step_driver_create(table, &r->drv);
  const size_t n = max_batch;
  unsigned m2 = 4096;
  while ((size_t)m2 < 2 * n) m2 <<= 1;
  r->m2 = m2;
  for (int p = 0; p < 2 && !rc; ++p) {
    rc = dmalloc(&r->ent[p], (size_t)m2 + 2);
    if (!rc) rp_fill_kernel<<<256, 256, 0, nullptr>>>(r->ent[p], (size_t)m2 + 2);
  }
  if (!rc) rc = dmalloc(&r->gcount, (size_t)2 * RP_MAX_WORLD);
  if (!rc && hipMemset(r->gcount, 0, (size_t)2 * RP_MAX_WORLD * sizeof(unsigned)) != hipSuccess) rc = hip_fail("memset");
  for (ASlot& sl : r->slots) {
    if (rc) break;
    sl.send.assign(r->world, 0); sl.recv.assign(r->world, 0);
    rc = dmalloc(&sl.owner_major, n);
    if (!rc) rc = dmalloc(&sl.lastpos, n);
    if (!rc) rc = dmalloc(&sl.pos2row, n);
    if (!rc) rc = dmalloc(&sl.d_counts, (size_t)2 * r->world);
    if (!rc && r->has_tr) { rc = dmalloc(&sl.recv_buf, n); if (!rc) sl.rcap = n; }
    if (!rc && hipHostMalloc(reinterpret_cast<void**>(&sl.h_counts), (size_t)2 * r->world * sizeof(int64_t), hipHostMallocDefault) != hipSuccess)
      rc = hip_fail("pinned allocation");
    for (hipEvent_t* e : {&sl.src_ev, &sl.done})
      if (!rc && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = hip_fail("event create");
  }
  for (hipEvent_t& e : r->marks)
    if (!rc && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = hip_fail("event create");
  if (!rc) rc = dmalloc(&r->vsend, n * r->row_bytes);
  if (r->has_tr) {
    if (!rc) rc = dmalloc(&r->rows_back, n * r->row_bytes);
    if (!rc) rc = ensure_served(r, n);
  } else {   // one rank: a rank's own send buffers are its receive buffers (nr == u <= max_batch)
    if (!rc) rc = dmalloc(&r->rows_served, n * r->row_bytes);
    r->vrecv = r->vsend; r->rows_back = r->rows_served; r->served_cap = n;
  }
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = hip_fail("synchronize");
  if (rc) { std::string keep = tfra::g_last_error; (void)tfra_assign_route_destroy(r); tfra::g_last_error = keep; return rc; }
  *out = r;
  return TFRA_OK;
}

int tfra_assign_route_feed(tfra_assign_route_t* r, size_t n, const int64_t* d_ids, int ids_ready, tfra_stream_t stream) {
  if (!r) return set_error(TFRA_ERR_INVALID, "assign_route_feed: null route");
  if (r->live >= MAX_LIVE) return set_error(TFRA_ERR_INVALID, "assign_route_feed: six batches are fed ahead already");
  if (n == 0 || n > r->max_n || !d_ids) return set_error(TFRA_ERR_INVALID, "assign_route_feed: 1 <= n <= max_batch ids expected");
  { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != r->device) { if (hipSetDevice(r->device) != hipSuccess) return hip_fail("hipSetDevice"); } }
  ASlot& sl = r->slots[r->tail];
  sl.wait_src = !ids_ready;   // the ids are still being produced on the caller's stream
  if (sl.wait_src && hipEventRecord(sl.src_ev, (hipStream_t)stream) != hipSuccess) return hip_fail("event record");
  sl.ids = d_ids; sl.n = n; sl.u = sl.nr = 0;
  int rc = issue_plan(r, sl);
  if (rc) return rc;
  r->tail = (r->tail + 1) % NS;
  r->live += 1;
  return TFRA_OK;
}

int tfra_assign_route_step(tfra_assign_route_t* r, void* d_rows_out, const void* default_row, const void* values_prev, tfra_stream_t stream) {
  if (!r) return set_error(TFRA_ERR_INVALID, "assign_route_step: null route");
  const int fed_ahead = r->live - (r->pending ? 1 : 0);
  if (fed_ahead <= 0) return set_error(TFRA_ERR_INVALID, "assign_route_step: no batch fed (tfra_assign_route_flush writes the pending batch back)");
  if (!d_rows_out || !default_row) return set_error(TFRA_ERR_INVALID, "assign_route_step: null buffer");
  if (r->pending && !values_prev) return set_error(TFRA_ERR_INVALID, "assign_route_step: the previous step's batch has not been written back: values_prev is null");
  { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != r->device) { if (hipSetDevice(r->device) != hipSuccess) return hip_fail("hipSetDevice"); } }
  hipStream_t s = (hipStream_t)stream;
  ASlot& cur = r->slots[r->look];
  int rc = ensure_routed(r, cur);
  if (rc) return rc;
  // the two batches behind it, if their ids have arrived (never forced: a missing look-ahead costs a plan launch, not a result)
  ASlot* nxt = fed_ahead >= 2 ? &r->slots[(r->look + 1) % NS] : nullptr;
  if (nxt && nxt->state != ST_ROUTED) nxt = nullptr;
  ASlot* nx2 = (nxt && fed_ahead >= 3) ? &r->slots[(r->look + 2) % NS] : nullptr;
  if (nx2 && nx2->state != ST_ROUTED) nx2 = nullptr;
  // the ahead stream is in order: waiting for the newest batch's ids covers the older ones (and every route plan before them)
  ASlot* newest = nx2 ? nx2 : (nxt ? nxt : &cur);
  if (!newest->main_waited && hipStreamWaitEvent(s, newest->ids_ev, 0) != hipSuccess) return hip_fail("event wait");
  cur.main_waited = true;
  if (nxt) nxt->main_waited = true;
  if (nx2) nx2->main_waited = true;
  ASlot* pv = r->pending ? &r->slots[r->head] : nullptr;
  if (pv) { rc = send_values(r, *pv, values_prev, s); if (rc) return rc; }
  const bool wb = pv && pv->nr > 0;   // this rank received rows to write back
  if (cur.nr) {
    const bool n1 = nxt && nxt->nr, n2 = n1 && nx2 && nx2->nr;
    rc = tfra_table_step_overlap(r->drv, cur.nr, (const int64_t*)cur.recv_ids, r->rows_served, nullptr, default_row, 0, wb ? r->vrecv : nullptr, nullptr,
                                 n1 ? nxt->nr : 0, n1 ? (const int64_t*)nxt->recv_ids : nullptr, n2 ? nx2->nr : 0,
                                 n2 ? (const int64_t*)nx2->recv_ids : nullptr, stream);
  } else if (wb) {
    rc = tfra_table_step_overlap_flush(r->drv, r->vrecv, nullptr, stream);   // nothing to look up here: the write-back alone
  }
  if (rc) return rc;
  rc = a2a(r, 0, r->rows_served, cur.recv, r->rows_back, cur.send, r->row_bytes, s);
  if (rc) return rc;
  rc = tfra_gather_rows(cur.n, r->row_bytes, r->rows_back, cur.pos2row, d_rows_out, stream);
  if (rc) return rc;
  if (pv) retire_head(r, s);
  cur.state = ST_LOOKED;
  r->pending = true;
  r->look = (r->look + 1) % NS;
  r->n_steps += 1;
  return advance_ahead(r);
}

int tfra_assign_route_flush(tfra_assign_route_t* r, const void* values_prev, tfra_stream_t stream) {
  if (!r) return set_error(TFRA_ERR_INVALID, "assign_route_flush: null route");
  if (!r->pending) return TFRA_OK;
  if (!values_prev) return set_error(TFRA_ERR_INVALID, "assign_route_flush: null values_prev");
  { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != r->device) { if (hipSetDevice(r->device) != hipSuccess) return hip_fail("hipSetDevice"); } }
  hipStream_t s = (hipStream_t)stream;
  ASlot& pv = r->slots[r->head];
  int rc = send_values(r, pv, values_prev, s);
  if (rc) return rc;
  if (pv.nr) { rc = tfra_table_step_overlap_flush(r->drv, r->vrecv, nullptr, stream); if (rc) return rc; }
  retire_head(r, s);
  return TFRA_OK;
}

// measurement: HIP events around the owner's step launch of each of the next `steps` steps (tfra_step_driver_time_kernels)
int tfra_assign_route_time_kernels(tfra_assign_route_t* r, size_t steps) {
  if (!r) return set_error(TFRA_ERR_INVALID, "assign_route_time_kernels: null route");
  return tfra_step_driver_time_kernels(r->drv, steps);
}
int tfra_assign_route_kernel_times(tfra_assign_route_t* r, double* step_kernel_us, size_t* steps) {
  if (!r) return set_error(TFRA_ERR_INVALID, "assign_route_kernel_times: null route");
  return tfra_step_driver_kernel_times(r->drv, step_kernel_us, nullptr, steps);
}

int tfra_assign_route_stats(const tfra_assign_route_t* r, uint64_t* out6) {
  if (!r || !out6) return set_error(TFRA_ERR_INVALID, "assign_route_stats: null argument");
  uint64_t ov = 0, sq = 0;
  int rc = tfra_step_driver_stats(r->drv, &ov, &sq, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  const ASlot& last = r->slots[(r->look + NS - 1) % NS];
  out6[0] = r->n_steps; out6[1] = r->n_stalls; out6[2] = ov; out6[3] = sq;
  out6[4] = r->n_steps ? last.u : 0; out6[5] = r->n_steps ? last.nr : 0;
  return TFRA_OK;
}

}  // extern "C"
