// The metric's step — lookup(B ids) + insert_or_assign(B ids, B rows) — on a table hash-sharded over the GPUs of one node.
// Reference shape: HvdAllToAllEmbedding / __alltoall_embedding_lookup__ (PY/shadow_embedding_ops.py:397-447: unique ->
// partition by owner -> alltoall(ids) -> local lookup -> alltoall(rows) -> stitch) and, for the write-back, the same route taken by
// the values (`Variable.upsert` partitions keys AND values by owner: PY/dynamic_embedding_variable.py:772-800).
//
// What is different here (MI355X-first, not the reference's call pattern):
//   * the OWNER's half of a step is the overlapped step of tfra_step_impl.h — lookup of the ids it serves for batch i+1 and
//     write-back of the rows it received for batch i in ONE launch (ids the two share are forwarded from the received rows);
//   * everything that depends on the ids alone runs AHEAD of the step, off its critical path, and is ONE launch per batch:
//     `routeplan_kernel` de-duplicates the batch (the distinct ids are what travels: a Zipf-1.2 batch of 131 072 ids has ~22 K),
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
// alone — identical on every rank.  No helper thread: the id-only half is one launch.
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

// ------------------------------------------------------------------------------------------- the id-only half: ONE launch
// routeplan_kernel: for a batch of n ids (<= 2^18)
//   keys_out[0 .. U)   the distinct ids grouped by owner (owner 0's first), order inside a group unspecified
//   lastpos[j]         the LAST position of keys_out[j] in the batch
//   pos2row[p]         the row j with keys_out[j] == ids[p]
//   d_counts[w]        distinct ids owned by rank w (int64: what the count exchange sends)
// Phases (blocks of 1024 threads, at most 128 of them, all co-resident — the kernel holds one grid-wide meeting):
//   A  equal ids of a block meet in an LDS hash table (compare-and-swap on the key, max on position + 1);
//   B  every distinct id of the block goes into a global open-addressing table (compare-and-swap; max on position + 1); the block
//      whose swap installed a key counts it under its owner (LDS), then draws the block's base per owner from the per-owner
//      counters (one returned add per block and owner) and ARRIVES;
//   C  while thread 0 waits for the other blocks' arrival, the block empties its share of the slots the PREVIOUS build used in
//      the other of the two tables (they alternate: no fill launch between builds);
//   D  all counts are final: prefix over the owners -> the row of every installed key = prefix[owner] + block base + index in
//      the block; the installer writes key / last position (final: every block's max came before its arrival) / slot and
//      publishes row + 1 in the entry; the other blocks holding the key poll that word (their installer is resident);
//   E  pos2row from the block's LDS table.
// The two sentinel key values (EMPTY_KEY, LOCKED_KEY) have slots of their own behind the table.
constexpr int RP_NT = 1024;
constexpr unsigned RP_MAX_WORLD = 64;
constexpr size_t RP_MAX_IDS = (size_t)1 << 18;
struct RpEnt { i64 key; unsigned pos1, row1; };   // key (EMPTY_KEY = free) | last position + 1 | row + 1 (0 = not yet known)
// control words of one use of one table (512 B): [0] blocks arrived, [1] distinct ids in total (left for the next build: the list
// it walks to empty this table), [32 .. 32 + world) distinct ids per owner
constexpr unsigned RP_CTL_WORDS = 128;

struct RpArgs {
  unsigned n, m2, world, mode, nblk;
  const i64* ids;
  RpEnt* ent; unsigned* slots; unsigned* ctl; unsigned* ctl_next_use;
  RpEnt* old_ent; const unsigned* old_slots; const unsigned* old_total;
  i64* keys_out; int* lastpos; int* pos2row; i64* d_counts;
  unsigned* err;   // pinned: meetings / polls that timed out (never seen; reported by the next call)
};

__device__ __forceinline__ int rp_owner_of(i64 key, unsigned num, unsigned mode) {   // default_partition_fn, PY/dynamic_embedding_variable.py:165-197
  if (mode == 0) return (int)((unsigned)(key & 0x7fffffff) % num);                    // (as tfra_partition: CUDA-build branch / CPU-build branch / opt-in mix)
  if (mode == 1) { i64 m = key % (i64)num; return (int)(m < 0 ? m + (i64)num : m); }
  return (int)__umul64hi(fmix64((u64)key), (u64)num);
}

template <int IPT>
__global__ __launch_bounds__(RP_NT) void routeplan_kernel(const RpArgs a) {
  constexpr unsigned LDSN = 2048u * IPT;   // slots of the block's LDS table: two per id
  constexpr int NR = 2 * IPT;
  __shared__ i64 s_key[LDSN];
  __shared__ unsigned s_pos[LDSN + 2], s_row[LDSN + 2];
  __shared__ unsigned s_own[RP_MAX_WORLD], s_obase[RP_MAX_WORLD], s_pref[RP_MAX_WORLD];
  const unsigned tid = threadIdx.x, bid = blockIdx.x, m2 = a.m2;
  const unsigned n_old = *a.old_total;
  for (unsigned i = tid; i < LDSN + 2; i += RP_NT) { if (i < LDSN) s_key[i] = EMPTY_KEY; s_pos[i] = 0; s_row[i] = 0; }
  if (tid < RP_MAX_WORLD) { s_own[tid] = 0; s_obase[tid] = 0; }
  if (bid == 0 && tid < RP_CTL_WORDS) a.ctl_next_use[tid] = 0;   // (its last reader — the other table's build after this table's previous use — is over)
  __syncthreads();
  // ---- A ------------------------------------------------------------------------------------------------------------
  unsigned gids[IPT], lds_slot[IPT];
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    gids[q] = (bid * IPT + q) * RP_NT + tid;
    lds_slot[q] = 0;
    if (gids[q] < a.n) {
      const i64 id = a.ids[gids[q]];
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
      atomicMax(&s_pos[slot], gids[q] + 1u);
      lds_slot[q] = slot;
    }
  }
  __syncthreads();
  // ---- B ------------------------------------------------------------------------------------------------------------
  // NR + 1 items per thread: the block's LDS slots tid, tid + 1024, ...; threads 0 and 1 also carry the two sentinel slots
  i64 mykey[NR + 1];
  unsigned myslot[NR + 1], myidx[NR + 1], myown[NR + 1], lslot[NR + 1];
  bool have[NR + 1], mine[NR + 1];
#pragma unroll
  for (int r = 0; r <= NR; ++r) {
    const bool sentinel = r == NR;
    lslot[r] = sentinel ? LDSN + (tid & 1u) : tid + (unsigned)r * RP_NT;
    have[r] = (!sentinel || tid < 2) && s_pos[lslot[r]] != 0;
    mine[r] = false; myidx[r] = 0; myown[r] = 0; myslot[r] = 0;
    mykey[r] = sentinel ? EMPTY_KEY + (i64)(tid & 1u) : s_key[lslot[r]];
    if (!have[r]) continue;
    const unsigned p1 = s_pos[lslot[r]];
    if (sentinel) {
      myslot[r] = m2 + (tid & 1u);
      const i64 w = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&a.ent[myslot[r]].key), (unsigned long long)EMPTY_KEY, 1ULL);
      mine[r] = w == EMPTY_KEY;
    } else {
      unsigned sl = (unsigned)(fmix64((u64)mykey[r]) >> 20) & (m2 - 1);
      for (;;) {
        const i64 w = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&a.ent[sl].key), (unsigned long long)EMPTY_KEY, (unsigned long long)mykey[r]);
        if (w == EMPTY_KEY) { mine[r] = true; break; }
        if (w == mykey[r]) break;
        sl = (sl + 1) & (m2 - 1);
      }
      myslot[r] = sl;
    }
    atomicMax(&a.ent[myslot[r]].pos1, p1);
    if (mine[r]) {
      myown[r] = (unsigned)rp_owner_of(mykey[r], a.world, a.mode);
      myidx[r] = atomicAdd(&s_own[myown[r]], 1u);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's max-es are acknowledged before the block arrives
  __syncthreads();
  if (tid < a.world && s_own[tid]) s_obase[tid] = atomicAdd(a.ctl + 32 + tid, s_own[tid]);   // (returned: performed when the value is here)
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- C: empty the slots the previous build used in the OTHER table ----------------------------------------------------
  for (unsigned i = bid * RP_NT + tid; i < n_old; i += a.nblk * RP_NT) {
    const unsigned sl = a.old_slots[i];
    *reinterpret_cast<uint4*>(a.old_ent + sl) = make_uint4(0u, 0x80000000u, 0u, 0u);   // {EMPTY_KEY, 0, 0}
  }
  // ---- the meeting ---------------------------------------------------------------------------------------------------
  if (tid == 0) {
    bool ok = false;
    for (unsigned it = 0; it < (1u << 24); ++it) {
      if (__hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.nblk) { ok = true; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    if (!ok) atomicAdd(a.err, 1u);
  }
  __syncthreads();
  // ---- D ------------------------------------------------------------------------------------------------------------
  if (tid < a.world) s_pref[tid] = __hip_atomic_load(a.ctl + 32 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (tid == 0) {
    unsigned run = 0;
    for (unsigned w = 0; w < a.world; ++w) {
      const unsigned c = s_pref[w];
      if (bid == 0) a.d_counts[w] = (i64)c;
      s_pref[w] = run;
      run += c;
    }
    if (bid == 0) a.ctl[1] = run;
  }
  __syncthreads();
  bool timed_out = false;
#pragma unroll
  for (int r = 0; r <= NR; ++r) {
    if (!have[r] || !mine[r]) continue;
    const unsigned row = s_pref[myown[r]] + s_obase[myown[r]] + myidx[r];
    a.keys_out[row] = mykey[r];
    a.slots[row] = myslot[r];
    a.lastpos[row] = (int)__hip_atomic_load(&a.ent[myslot[r]].pos1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1;
    __hip_atomic_store(&a.ent[myslot[r]].row1, row + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_row[lslot[r]] = row + 1u;
  }
#pragma unroll
  for (int r = 0; r <= NR; ++r) {
    if (!have[r] || mine[r]) continue;
    unsigned v = 0;
    for (unsigned it = 0; !v && it < (1u << 24); ++it) {
      v = __hip_atomic_load(&a.ent[myslot[r]].row1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!v) __builtin_amdgcn_s_sleep(2);
    }
    timed_out |= v == 0u;
    s_row[lslot[r]] = v;
  }
  if (timed_out) atomicAdd(a.err, 1u);
  __syncthreads();
  // ---- E ------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int q = 0; q < IPT; ++q)
    if (gids[q] < a.n) a.pos2row[gids[q]] = (int)s_row[lds_slot[q]] - 1;
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
constexpr int NS = 8;   // batches in flight: the one being written back, the one looked up, up to five fed ahead, one spare
// A batch moves through these stages (who issues what: everything the calling thread, in call order):
//   FED      routeplan_kernel                                                                [side stream]
//   COUNTED  alltoall of the per-owner counts, copy to pinned memory                         [coll stream]
//   ROUTED   split sizes read on the host, alltoall of the distinct ids                      [coll stream]
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
  i64* recv_ids = nullptr;      // [rcap] the ids this rank serves for the batch, source-major
  size_t rcap = 0;
  hipEvent_t src_ev = nullptr, plan_ev = nullptr, counts_ev = nullptr, ids_ev = nullptr, done = nullptr;
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
  hipStream_t side = nullptr, coll = nullptr;
  // routeplan scratch: two tables that alternate and empty each other, each with two control blocks that alternate per use
  RpEnt* ent[2] = {nullptr, nullptr};
  unsigned* uslots[2] = {nullptr, nullptr};
  unsigned* ctl = nullptr;        // [2 tables][2 uses][RP_CTL_WORDS]
  unsigned m2 = 0, builds = 0, uses[2] = {0, 0};
  unsigned* h_err = nullptr;      // pinned
  ASlot slots[NS];
  int head = 0, look = 0, tail = 0;   // oldest live slot (looked up, not yet written back — or == look), next to look up, next free
  int live = 0;                        // slots between head and tail
  bool pending = false;                // slots[head] has been looked up and awaits its write-back
  // critical-path buffers (caller's stream only)
  unsigned char* vsend = nullptr; unsigned char* rows_back = nullptr;            // [max_n rows]
  unsigned char* vrecv = nullptr; unsigned char* rows_served = nullptr; size_t served_cap = 0;   // [served_cap rows]
  std::vector<size_t> sb, rb;
  unsigned long long n_steps = 0, n_stalls = 0;
};

namespace {

int a2a(tfra_assign_route* r, int channel, const void* send, const std::vector<size_t>& sc, void* recv, const std::vector<size_t>& rc, size_t elem,
        hipStream_t s) {
  if (!r->has_tr) {   // one rank: what it sends is what it receives
    if (sc[0] && hipMemcpyAsync(recv, send, sc[0] * elem, hipMemcpyDeviceToDevice, s) != hipSuccess) return hip_fail("local copy");
    return TFRA_OK;
  }
  for (int i = 0; i < r->world; ++i) { r->sb[i] = sc[i] * elem; r->rb[i] = rc[i] * elem; }
  return r->tr.alltoallv(r->tr.ctx, channel, send, r->sb.data(), recv, r->rb.data(), (tfra_stream_t)s);
}

int ensure_served(tfra_assign_route* r, size_t nr) {
  if (nr <= r->served_cap) return TFRA_OK;
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

int check_err_word(tfra_assign_route* r) {
  if (r->h_err && __atomic_load_n(r->h_err, __ATOMIC_RELAXED)) {
    __atomic_store_n(r->h_err, 0u, __ATOMIC_RELAXED);
    return set_error(TFRA_ERR_HIP, "assign route: a route-plan launch timed out waiting for its other blocks (the GPU did not keep <= 128 blocks resident)");
  }
  return TFRA_OK;
}

// FED: the one launch of the id-only half
int issue_plan(tfra_assign_route* r, ASlot& sl) {
  hipStream_t side = r->side;
  if (sl.wait_src && hipStreamWaitEvent(side, sl.src_ev, 0) != hipSuccess) return hip_fail("event wait");
  // the slot's buffers were last read by the step that wrote its previous batch back
  if (sl.done_recorded && hipEventQuery(sl.done) != hipSuccess && hipStreamWaitEvent(side, sl.done, 0) != hipSuccess) return hip_fail("event wait");
  const unsigned p = r->builds & 1u;
  r->builds += 1;
  const unsigned use = ++r->uses[p];
  RpArgs a{};
  a.n = (unsigned)sl.n; a.m2 = r->m2; a.world = (unsigned)r->world; a.mode = (unsigned)r->mode;
  const int ipt = sl.n <= (size_t)128 * RP_NT ? 1 : 2;
  a.nblk = (unsigned)((sl.n + (size_t)RP_NT * ipt - 1) / ((size_t)RP_NT * ipt));
  a.ids = (const i64*)sl.ids;
  a.ent = r->ent[p]; a.slots = r->uslots[p];
  a.ctl = r->ctl + (size_t)(2 * p + (use & 1u)) * RP_CTL_WORDS;
  a.ctl_next_use = r->ctl + (size_t)(2 * p + ((use + 1) & 1u)) * RP_CTL_WORDS;
  a.old_ent = r->ent[p ^ 1u]; a.old_slots = r->uslots[p ^ 1u];
  a.old_total = r->ctl + (size_t)(2 * (p ^ 1u) + (r->uses[p ^ 1u] & 1u)) * RP_CTL_WORDS + 1;   // (never used yet: a zeroed word)
  a.keys_out = sl.owner_major; a.lastpos = sl.lastpos; a.pos2row = sl.pos2row; a.d_counts = sl.d_counts;
  a.err = r->h_err;
  if (ipt == 1) routeplan_kernel<1><<<a.nblk, RP_NT, 0, side>>>(a);
  else routeplan_kernel<2><<<a.nblk, RP_NT, 0, side>>>(a);
  if (hipGetLastError() != hipSuccess) return hip_fail("route-plan launch failed");
  if (hipEventRecord(sl.plan_ev, side) != hipSuccess) return hip_fail("event record");
  sl.state = ST_FED;
  return TFRA_OK;
}

// FED -> COUNTED
int issue_counts(tfra_assign_route* r, ASlot& sl) {
  hipStream_t c = r->coll;
  if (hipStreamWaitEvent(c, sl.plan_ev, 0) != hipSuccess) return hip_fail("event wait");
  if (r->has_tr) {
    for (int i = 0; i < r->world; ++i) r->sb[i] = r->rb[i] = sizeof(int64_t);
    int rc = r->tr.alltoallv(r->tr.ctx, 1, sl.d_counts, r->sb.data(), sl.d_counts + r->world, r->rb.data(), (tfra_stream_t)c);
    if (rc) return rc;
  } else if (hipMemcpyAsync(sl.d_counts + 1, sl.d_counts, sizeof(int64_t), hipMemcpyDeviceToDevice, c) != hipSuccess) {
    return hip_fail("local copy");
  }
  if (hipMemcpyAsync(sl.h_counts, sl.d_counts, (size_t)2 * r->world * sizeof(int64_t), hipMemcpyDeviceToHost, c) != hipSuccess ||
      hipEventRecord(sl.counts_ev, c) != hipSuccess)
    return hip_fail("split sizes copy");
  sl.state = ST_COUNTED;
  return TFRA_OK;
}

// COUNTED -> ROUTED: the ONE host read of a batch (the split sizes, as hvd.alltoall(ids, splits) needs them too), then the ids
int issue_ids(tfra_assign_route* r, ASlot& sl) {
  if (hipEventQuery(sl.counts_ev) != hipSuccess) {
    r->n_stalls += 1;
    if (hipEventSynchronize(sl.counts_ev) != hipSuccess) return hip_fail("waiting for the split sizes");
  }
  int rc = check_err_word(r);
  if (rc) return rc;
  size_t u = 0, nr = 0;
  for (int i = 0; i < r->world; ++i) {
    sl.send[i] = (size_t)sl.h_counts[i]; sl.recv[i] = (size_t)sl.h_counts[r->world + i];
    u += sl.send[i]; nr += sl.recv[i];
  }
  if (u > sl.n || u == 0) return set_error(TFRA_ERR_INVALID, "assign route: impossible split sizes (ranks out of step?)");
  if (nr > RP_MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "assign route: a rank serves at most 2^18 ids per batch");
  sl.u = u; sl.nr = nr;
  if (nr > sl.rcap) {
    if (hipDeviceSynchronize() != hipSuccess) return hip_fail("synchronize before growing");
    (void)hipFree(sl.recv_ids); sl.recv_ids = nullptr; sl.rcap = 0;
    const size_t cap = std::min(RP_MAX_IDS, nr + nr / 4 + 1024);
    rc = dmalloc(&sl.recv_ids, cap);
    if (rc) return rc;
    sl.rcap = cap;
  }
  rc = ensure_served(r, nr);
  if (rc) return rc;
  rc = a2a(r, 1, sl.owner_major, sl.send, sl.recv_ids, sl.recv, sizeof(int64_t), r->coll);
  if (rc) return rc;
  if (hipEventRecord(sl.ids_ev, r->coll) != hipSuccess) return hip_fail("event record");
  sl.main_waited = false;
  sl.state = ST_ROUTED;
  return TFRA_OK;
}

int ensure_routed(tfra_assign_route* r, ASlot& sl) {
  int rc = TFRA_OK;
  if (sl.state == ST_FED) rc = issue_counts(r, sl);
  if (!rc && sl.state == ST_COUNTED) rc = issue_ids(r, sl);
  return rc;
}

// after a step: every batch fed ahead moves one stage.  Depends on the call sequence alone (the same on every rank).
int advance_ahead(tfra_assign_route* r) {
  for (int k = 0, i = r->head; k < r->live; ++k, i = (i + 1) % NS) {
    ASlot& sl = r->slots[i];
    if (sl.state == ST_COUNTED) { int rc = issue_ids(r, sl); if (rc) return rc; }
    else if (sl.state == ST_FED) { int rc = issue_counts(r, sl); if (rc) return rc; }
  }
  return TFRA_OK;
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
    (void)hipFree(sl.owner_major); (void)hipFree(sl.lastpos); (void)hipFree(sl.pos2row); (void)hipFree(sl.d_counts); (void)hipFree(sl.recv_ids);
    if (sl.h_counts) (void)hipHostFree(sl.h_counts);
    for (hipEvent_t e : {sl.src_ev, sl.plan_ev, sl.counts_ev, sl.ids_ev, sl.done}) if (e) (void)hipEventDestroy(e);
  }
  for (int p = 0; p < 2; ++p) { (void)hipFree(r->ent[p]); (void)hipFree(r->uslots[p]); }
  (void)hipFree(r->ctl);
  if (r->h_err) (void)hipHostFree(r->h_err);
  (void)hipFree(r->vsend); (void)hipFree(r->rows_back); (void)hipFree(r->vrecv); (void)hipFree(r->rows_served);
  if (r->side) (void)hipStreamDestroy(r->side);
  if (r->coll) (void)hipStreamDestroy(r->coll);
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
  r->sb.resize(r->world); r->rb.resize(r->world);
  int rc = hipSetDevice(r->device) == hipSuccess ? TFRA_OK : hip_fail("hipSetDevice");
  if (!rc && (hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking) != hipSuccess ||
              hipStreamCreateWithFlags(&r->coll, hipStreamNonBlocking) != hipSuccess)) rc = hip_fail("stream create");
  if (!rc) rc = tfra_step_driver_create(table, &r->drv);
  const size_t n = max_batch;
  unsigned m2 = 4096;
  while ((size_t)m2 < 2 * n) m2 <<= 1;
  r->m2 = m2;
  for (int p = 0; p < 2 && !rc; ++p) {
    rc = dmalloc(&r->ent[p], (size_t)m2 + 2);
    if (!rc) rc = dmalloc(&r->uslots[p], n);
    if (!rc) rp_fill_kernel<<<256, 256, 0, nullptr>>>(r->ent[p], (size_t)m2 + 2);
  }
  if (!rc) rc = dmalloc(&r->ctl, (size_t)4 * RP_CTL_WORDS);
  if (!rc && hipMemset(r->ctl, 0, (size_t)4 * RP_CTL_WORDS * sizeof(unsigned)) != hipSuccess) rc = hip_fail("memset");
  if (!rc && hipHostMalloc(reinterpret_cast<void**>(&r->h_err), 64, hipHostMallocDefault) != hipSuccess) { r->h_err = nullptr; rc = hip_fail("pinned allocation"); }
  if (!rc) *r->h_err = 0;
  for (ASlot& sl : r->slots) {
    if (rc) break;
    sl.send.assign(r->world, 0); sl.recv.assign(r->world, 0);
    rc = dmalloc(&sl.owner_major, n);
    if (!rc) rc = dmalloc(&sl.lastpos, n);
    if (!rc) rc = dmalloc(&sl.pos2row, n);
    if (!rc) rc = dmalloc(&sl.d_counts, (size_t)2 * r->world);
    if (!rc) rc = dmalloc(&sl.recv_ids, n);
    if (!rc) sl.rcap = n;
    if (!rc && hipHostMalloc(reinterpret_cast<void**>(&sl.h_counts), (size_t)2 * r->world * sizeof(int64_t), hipHostMallocDefault) != hipSuccess)
      rc = hip_fail("pinned allocation");
    for (hipEvent_t* e : {&sl.src_ev, &sl.plan_ev, &sl.counts_ev, &sl.ids_ev, &sl.done})
      if (!rc && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = hip_fail("event create");
  }
  if (!rc) rc = dmalloc(&r->vsend, n * r->row_bytes);
  if (!rc) rc = dmalloc(&r->rows_back, n * r->row_bytes);
  if (!rc) rc = ensure_served(r, n);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = hip_fail("synchronize");
  if (rc) { std::string keep = tfra::g_last_error; (void)tfra_assign_route_destroy(r); tfra::g_last_error = keep; return rc; }
  *out = r;
  return TFRA_OK;
}

int tfra_assign_route_feed(tfra_assign_route_t* r, size_t n, const int64_t* d_ids, int ids_ready, tfra_stream_t stream) {
  if (!r) return set_error(TFRA_ERR_INVALID, "assign_route_feed: null route");
  if (r->live >= NS - 1) return set_error(TFRA_ERR_INVALID, "assign_route_feed: six batches are fed ahead already");
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
  // the coll stream is in order: waiting for the newest batch's ids covers the older ones (and their route plans)
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
