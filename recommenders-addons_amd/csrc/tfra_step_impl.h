// The OVERLAPPED STEP: lookup of batch i+1 in the same launch as the write-back of batch i.  Included twice by tfra_csr.hip
// (device part inside its anonymous namespace, host part at its end): it uses that file's ownership pass, plan kernels and
// plan object as they are.
//
// Reference semantics kept (hkv_hashtable_op_gpu.cu.cc:192-213,256-267: Insert exclusive, Find shared => lookup i+1 sees
// update i; lookup_table_op_hkv.h:522-537 upsert, :719-756 get with default fill): the results are those of
//     lookup(ids_0); insert_or_assign(ids_0, values_0); lookup(ids_1); insert_or_assign(ids_1, values_1); ...
// executed one after the other.
//
// Why it may overlap.  What keeps lookup(i+1) behind write-back(i) is the keys the two share — and for exactly those the row
// the lookup must return is known without the table: values_i[last position of the key in batch i], which the SET plan of
// batch i (an open-addressing table key -> last position, built one batch ahead) gives in one probe.  So:
//   * lookup(i+1) probes plan(i) for every id; a hit is served from values_i (STORE-TO-LOAD FORWARDING), everything else
//     from the table.  Keys that are NOT in batch i keep their row through write-back(i) — unless write-back(i) EVICTS them;
//   * write-back(i) therefore probes plan(i+1) for every victim it is about to replace: a victim the next lookup looks for
//     is not evicted in the pass; the new key goes to the remainder pass, which runs AFTER the lookup, evicts then, and
//     rewrites the lookup's output rows of that victim with the default row (what a lookup after the eviction returns).
//     (~160 evictions per step on the metric's configuration, practically none of them of a key of the next batch.)
//   * everything else the pass does to a bucket (rows of keys of batch i, free-slot inserts, flags) is invisible to a lookup of
//     OTHER keys: slots are 8-byte words, flags are monotone.
// Forwarding needs every key of batch i to END UP in the table (a key the table does not admit must read as absent):
// LRU-type scores on a bounded table at capacity — the metric's configuration.  Anything else takes the sequential fallback
// inside the same entry point.
//
// One launch, three roles (block-uniform branches, no cross-role synchronisation):
//   blocks [0, P)          PLAN   the SET plan of batch i+2 (setplan_kernel's algorithm, 1024 ids per 256-thread block)
//   blocks [P, P+O)        OWN    write-back(i): own_batch16 over plan(i)'s keys, victims checked against plan(i+1)
//   blocks [P+O, P+O+F)    FIND   lookup(i+1) with forwarding from plan(i) / values_i
// then step_rest_kernel: the keys the pass left over (lost claims, deferred evictions) with the locked protocol + the output
// corrections.  Two launches per step on ONE stream, no events, no host in the loop: the sequence can be enqueued many steps
// ahead (tfra_table_steps_overlap) or captured into a graph.

#ifdef TFRA_STEP_DEVICE_PART

struct StepArgs {
  OwnArgs own;                 // write-back of the PREVIOUS batch (SRC_SET: own.ks = its plan); own_blocks == 0: none pending
  OwnCtrs* ctr;                // its left-over counters
  unsigned own_gen;
  unsigned* progress;          // pinned: [0] step, [1] distinct keys of this write-back (sizes the next one's grid)
  unsigned progress_val;
  SetProbe fwd;                // the previous batch's plan: ids found here are served from own.vals
  SetProbe nxt;                // THIS batch's plan: victims found here are not evicted in the pass
  // lookup of this batch
  unsigned n;
  const i64* ids;
  unsigned char* out;
  uint8_t* exists;
  const unsigned char* defaults;
  int full;
  // plan role: the plan of the NEXT batch
  unsigned n_plan;
  const i64* ids_plan;
  SetTab pcur, pold;
  unsigned* next_use_count;
  unsigned plan_m2;
  unsigned plan_blocks, own_blocks;
  unsigned* stat;              // [0] evictions the pass deferred, [1] victims the remainder noted, [2] output rows corrected
};

// ---- PLAN role: setplan_kernel<false> for 256 threads per 1024 ids -------------------------------------------------
constexpr unsigned SPK_IDS = 1024, SPK_LDS = 2048, SPK_PER = SPK_LDS / 256;
struct PlanLds {
  i64 key[SPK_LDS];
  unsigned pos[SPK_LDS + 2];
  unsigned n, base;
};

__device__ __forceinline__ void plan_role(const StepArgs& a, unsigned blk, PlanLds& L) {
  const unsigned tid = threadIdx.x;
  const SetTab& cur = a.pcur;
  const SetTab& old = a.pold;
  const unsigned m2 = a.plan_m2;
  const unsigned n_old = *old.count;
  if (blk == 0 && tid == 0) *a.next_use_count = 0;
  for (unsigned i = tid; i < SPK_LDS + 2; i += 256) { if (i < SPK_LDS) L.key[i] = EMPTY_KEY; L.pos[i] = 0; }
  if (tid == 0) L.n = 0;
  __syncthreads();
  // A: equal ids of the block meet in LDS (4 ids per thread, coalesced)
  i64 id[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned g = blk * SPK_IDS + (unsigned)r * 256u + tid;
    id[r] = g < a.n_plan ? a.ids_plan[g] : 0;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned g = blk * SPK_IDS + (unsigned)r * 256u + tid;
    if (g >= a.n_plan) continue;
    unsigned slot;
    if (is_reserved_key(id[r])) slot = SPK_LDS + (unsigned)reserved_index(id[r]);
    else {
      slot = (unsigned)(fmix64((u64)id[r]) >> 41) & (SPK_LDS - 1);
      for (;;) {
        const i64 was = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&L.key[slot]), (unsigned long long)EMPTY_KEY, (unsigned long long)id[r]);
        if (was == EMPTY_KEY || was == id[r]) break;
        slot = (slot + 1) & (SPK_LDS - 1);
      }
    }
    atomicMax(&L.pos[slot], g + 1u);
  }
  __syncthreads();
  // B: the block's distinct ids into the global table; the first probes of a thread's 8 slots travel together
  i64 mykey[SPK_PER], was[SPK_PER];
  unsigned myslot[SPK_PER], p1[SPK_PER], myidx[SPK_PER];
  bool mine[SPK_PER];
#pragma unroll
  for (int r = 0; r < (int)SPK_PER; ++r) {
    const unsigned s = tid + (unsigned)r * 256u;
    mykey[r] = L.key[s];
    p1[r] = L.pos[s];
    myslot[r] = (unsigned)(fmix64((u64)mykey[r]) >> 20) & (m2 - 1);
    was[r] = 0;
    mine[r] = false;
    if (p1[r]) was[r] = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&cur.ent[myslot[r]].key), (unsigned long long)EMPTY_KEY, (unsigned long long)mykey[r]);
  }
#pragma unroll
  for (int r = 0; r < (int)SPK_PER; ++r) {
    if (p1[r]) {
      for (;;) {
        if (was[r] == EMPTY_KEY) { mine[r] = true; break; }
        if (was[r] == mykey[r]) break;
        myslot[r] = (myslot[r] + 1) & (m2 - 1);
        was[r] = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&cur.ent[myslot[r]].key), (unsigned long long)EMPTY_KEY, (unsigned long long)mykey[r]);
      }
      atomicMax(&cur.ent[myslot[r]].pos1, p1[r]);
    }
    myidx[r] = mine[r] ? atomicAdd(&L.n, 1u) : 0u;
  }
  if (tid < 2 && L.pos[SPK_LDS + tid] != 0) {   // a sentinel key value occurred in this block
    const unsigned sl = m2 + tid;
    const i64 w = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&cur.ent[sl].key), (unsigned long long)EMPTY_KEY, 1ULL);
    atomicMax(&cur.ent[sl].pos1, L.pos[SPK_LDS + tid]);
    if (w == EMPTY_KEY) {
      const unsigned at = atomicAdd(cur.count, 1u);
      cur.ukeys[at] = EMPTY_KEY + (i64)tid;
      cur.uslot[at] = sl;
    }
  }
  __syncthreads();
  if (tid == 0) L.base = L.n ? atomicAdd(cur.count, L.n) : 0u;
  // C: empty the slots the previous build used in the OTHER table
  for (unsigned i = blk * 256u + tid; i < n_old; i += a.plan_blocks * 256u)
    *reinterpret_cast<uint4*>(old.ent + old.uslot[i]) = make_uint4(0u, 0x80000000u, 0u, 0u);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < (int)SPK_PER; ++r) {
    if (!mine[r]) continue;
    cur.ukeys[L.base + myidx[r]] = mykey[r];
    cur.uslot[L.base + myidx[r]] = myslot[r];
  }
}

// ---- FIND role: find_kernel<16, 4, WT, PF1> + forwarding -----------------------------------------------------------
// Lane j (and j+16, j+32, j+48) holds key j of the wave's 16 and hashes it; for the plan probe the FOUR replicas of a key
// read four consecutive entries of its chain (one 16-B load per lane, all 64 lanes busy, no redundancy); the table probe is
// find_kernel's (both home buckets' lines in flight).  One wait for everything, then rows.
__device__ __forceinline__ void find_fwd_role(const StepArgs& a, unsigned blk) {
  constexpr int U = 4;
  const TableView& v = a.own.v;
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, grp = lane >> 4;
  const unsigned wave = blk * 4u + (threadIdx.x >> 6);
  const unsigned base = wave * 16u;
  if (base >= a.n) return;
  const unsigned last = a.n - 1;
  const i64 kreg = a.ids[min(base + (unsigned)sub, last)];
  u64 hreg;
  const unsigned b0reg = (unsigned)bucket0(kreg, v.nb, hreg);
  const unsigned b1reg = (unsigned)bucket1(hreg, b0reg, v.nb);
  const bool resv = is_reserved_key(kreg);
  const unsigned home = set_home(a.fwd, kreg, hreg);
  const unsigned eidx = resv ? home + (unsigned)grp : (home + (unsigned)grp) & (a.fwd.m2 - 1);
  uint4 e = *reinterpret_cast<const uint4*>(a.fwd.ent + eidx);
  i64 key[U], k0[U], k1[U];
  unsigned b0[U], b1[U], idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int j = u * 4 + grp;
    key[u] = shfl_i64(kreg, j);
    b0[u] = (unsigned)__shfl((int)b0reg, j);
    b1[u] = (unsigned)__shfl((int)b1reg, j);
    idx[u] = min(base + (unsigned)j, last);
    k0[u] = key_line(v, b0[u])[sub];
    k1[u] = key_line(v, b1[u])[sub];
  }
  keep_live(e.x, e.y, e.z, e.w);
  keep_live(k0[0], k0[1], k0[2], k0[3]);
  keep_live(k1[0], k1[1], k1[2], k1[3]);
  // the plan probe, per key: a match in any of the four entries is the key; none and no EMPTY among them: go on (rare)
  const i64 ekey = (i64)(((u64)e.y << 32) | e.x);
  const bool match = resv ? (grp == 0 && ekey != EMPTY_KEY) : ekey == kreg;
  unsigned p1 = match ? e.z : 0u;
  unsigned stop = (ekey == EMPTY_KEY || resv) ? 1u : 0u;
  p1 |= (unsigned)__shfl_xor((int)p1, 16); p1 |= (unsigned)__shfl_xor((int)p1, 32);
  stop |= (unsigned)__shfl_xor((int)stop, 16); stop |= (unsigned)__shfl_xor((int)stop, 32);
  if (!p1 && !stop) {
    for (unsigned t = 4; t < a.fwd.m2; ++t) {
      const SetEnt* q = a.fwd.ent + ((home + t) & (a.fwd.m2 - 1));
      const i64 k = q->key;
      if (k == kreg) { p1 = q->pos1; break; }
      if (k == EMPTY_KEY) break;
    }
  }
  const unsigned char* src[U];
  unsigned char* dst[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned fw = (unsigned)__shfl((int)p1, u * 4 + grp);
    i64 word = 0;
    if (!fw) word = probe_find_word(v, key[u], b0[u], b1[u], k0[u], sub, gshift, &k1[u]);
    if (a.exists && sub == 0) a.exists[idx[u]] = fw != 0 || word >= 0;
    src[u] = fw ? a.own.vals + (u64)(fw - 1u) * (u64)v.field_bytes
                : (word >= 0 ? word_row_ptr(v, (u64)word) : a.defaults + (a.full ? (u64)idx[u] * (u64)v.field_bytes : 0));
    dst[u] = a.out + (u64)idx[u] * (u64)v.field_bytes;
  }
  for (unsigned off = sub * 16; off < v.field_bytes; off += 256) {
    uint4 tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) tmp[u] = *reinterpret_cast<const uint4*>(src[u] + off);
    keep_live(tmp[0], tmp[1], tmp[2], tmp[3]);
#pragma unroll
    for (int u = 0; u < U; ++u) store_wt16(dst[u] + off, tmp[u]);
  }
}

// ---- OWN role: upsert_own_kernel<16, SIMPLE, SRC_SET, U> with the victim check ---------------------------------------
template <bool SIMPLE, int U>
__device__ __forceinline__ void own_role(const StepArgs& a, unsigned blk) {
  const OwnArgs& o = a.own;
  const int lane = threadIdx.x & 63;
  const unsigned total = o.ks.d_counts[0] + o.ks.d_counts[1];
  const unsigned nwaves = a.own_blocks * 4u;
  const unsigned wave = blk * 4u + (threadIdx.x >> 6);
  int fresh = 0;
  if (blk == 0 && threadIdx.x == 0 && a.progress) {
    __hip_atomic_store(a.progress, a.progress_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.progress + 1, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const OwnFlags fl = own_setup<SIMPLE>(o);
  for (unsigned wbase = wave * (4 * U); wbase < total; wbase += nwaves * (4 * U)) {
    const unsigned i = wbase + (unsigned)(lane & 15);
    own_batch16<16, SIMPLE, SRC_SET, U, true>(o, fl, min(i, total - 1), (lane & 15) < 4 * U && i < total, a.own_gen, &a.ctr->n_a, lane, fresh, &a.nxt, a.stat);
  }
  for (int off = 32; off > 0; off >>= 1) fresh += __shfl_xor(fresh, off);
  if (lane == 0 && fresh) size_add(o.v, wave, fresh);
}

// TIMING (tuning builds only): every block notes its start and end on the device clock; per launch and role the earliest start
// and the latest end are kept in a.stat (64-bit words behind the three counters): when does each role of a launch run?
__device__ __forceinline__ void role_stamp(const StepArgs& a, int role, u64 t0) {
  __syncthreads();
  if (threadIdx.x == 0) {
    u64* w = reinterpret_cast<u64*>(a.stat + 16) + ((size_t)(a.progress_val & 63u) * 3 + role) * 2;
    atomicMin(w, t0);
    atomicMax(w + 1, (u64)wall_clock64());
  }
}
template <bool SIMPLE, int U, bool PLAN, int MINW, bool TIMING = false>
__global__ __launch_bounds__(256, MINW) void step_kernel(const StepArgs a) {
  unsigned b = blockIdx.x;
  const u64 t0 = TIMING ? (u64)wall_clock64() : 0;
  if constexpr (PLAN) {
    __shared__ PlanLds L;
    if (b < a.plan_blocks) { plan_role(a, b, L); if (TIMING) role_stamp(a, 0, t0); return; }
    b -= a.plan_blocks;
  }
  if (b < a.own_blocks) { own_role<SIMPLE, U>(a, b); if (TIMING) role_stamp(a, 1, t0); return; }
  find_fwd_role(a, b - a.own_blocks);
  if (TIMING) role_stamp(a, 2, t0);
}

// ---- the remainder of a step: left-over keys of the pass + corrections of the lookup's output ------------------------
// upsert_rest_kernel<16, SRC_SET> over the pass's item list (the flags of all keys when the list overflowed).  Runs AFTER the
// launch that held the lookup, so an eviction here may hit a key that lookup has just returned a row for: every victim that is
// one of this batch's ids (plan `nxt`) is noted by the block.  When a block has noted any, it waits until EVERY block of the
// launch has finished its items (one arrival counter; blocks without such victims only arrive and leave — the grid is at most
// 512 blocks, all resident), looks the victims up again and, for those that are absent now, rewrites the lookup's output
// rows with the default row and clears their exists flags: what a lookup behind the write-back returns.  (A victim that is
// present again was a key of the previous batch whose own left-over write came later in this kernel.)
constexpr unsigned PATCH_CAP = 64;
__global__ __launch_bounds__(256) void step_rest_kernel(const StepArgs a, unsigned* zero4) {
  __shared__ i64 s_patch[PATCH_CAP];
  __shared__ unsigned s_np;
  __shared__ unsigned char s_absent[PATCH_CAP];
  const OwnArgs& o = a.own;
  const unsigned* slow_ctr = &a.ctr->n_a;
  unsigned* arrived = &a.ctr->spare[0];
  const unsigned gi = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (threadIdx.x == 0) s_np = 0;
  const OwnItem* it0 = o.items + (gi < o.item_cap ? gi : 0u);
  const uint4 f0 = reinterpret_cast<const uint4*>(it0)[0], f1 = reinterpret_cast<const uint4*>(it0)[1];
  const unsigned counted = *slow_ctr;
  if (zero4 && blockIdx.x == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0;   // last kernel of this use: arm the next use's counters
  if (counted == 0) return;
  __syncthreads();
  const bool listed = counted <= o.item_cap;
  const unsigned n = listed ? counted : o.ks.d_counts[0] + o.ks.d_counts[1];
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const unsigned ngroups = (gridDim.x * blockDim.x) >> 4;
  int fresh = 0, failed = 0;
  for (unsigned i = gi; i < n; i += ngroups) {
    i64 vk = EMPTY_KEY;
    if (listed) {
      uint4 w0 = f0, w1 = f1;
      if (i != gi) {
        w0 = reinterpret_cast<const uint4*>(o.items + i)[0];
        w1 = reinterpret_cast<const uint4*>(o.items + i)[1];
      }
      const i64 key = (i64)(((u64)w0.y << 32) | w0.x);
      locked_upsert_kv<16>(o.v, o.vals, key, w0.z, ((u64)w1.y << 32) | w1.x, o.ai, o.sp, sub, gshift, fresh, failed, w1.z != 0, w1.w, &vk);
      if (sub == 0) o.dflag[w0.w] = 0;
    } else {
      if (o.dflag[i] != 4) continue;
      const uint2 pc = set_pc(o.ks.sent + o.ks.uslot[i]);
      locked_upsert_kv<16>(o.v, o.vals, o.ks.ukeys[i], pc.x - 1, 1, o.ai, o.sp, sub, gshift, fresh, failed, false, 0, &vk);
      if (sub == 0) o.dflag[i] = 0;
    }
    if (vk != EMPTY_KEY && vk != LOCKED_KEY && a.n && set_contains_group(a.nxt, vk, sub, gshift)) {
      if (sub == 0) {
        const unsigned at = atomicAdd(&s_np, 1u);
        atomicAdd(a.stat + 1, 1u);
        if (at < PATCH_CAP) s_patch[at] = vk;
        else atomicAdd(o.v.err_count, 1u);   // (64 such victims in one block's share: reported by check_errors, never silent)
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) { fresh += __shfl_xor(fresh, off); failed += __shfl_xor(failed, off); }
  if (lane == 0) {
    if (fresh) size_add(o.v, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, fresh);
    if (failed) atomicAdd(o.v.err_count, (unsigned)failed);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's table stores (write-through / agent-scope) are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned np = min(s_np, PATCH_CAP);
  if (np == 0) return;
  if (threadIdx.x == 0) {
    bool ok = false;
    for (unsigned it = 0; it < (1u << 22) && !ok; ++it) {
      ok = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x;
      if (!ok) __builtin_amdgcn_s_sleep(16);
    }
    if (!ok) atomicAdd(o.v.err_count, 1u);   // (never seen: every block arrives without waiting for anything)
  }
  __syncthreads();
  // which of the noted victims are absent now?  (16 groups, one victim each per round; coherent loads)
  for (unsigned q = threadIdx.x >> 4; q < np; q += 16) {
    const i64 row = probe_find<true>(o.v, s_patch[q], sub, gshift);
    if (sub == 0) s_absent[q] = row < 0 ? 1 : 0;
  }
  __syncthreads();
  // those, against every id of the batch (rare: a scan of the ids by one block)
  for (unsigned p = threadIdx.x; p < a.n; p += blockDim.x) {
    const i64 id = a.ids[p];
    bool hit = false;
    for (unsigned q = 0; q < np; ++q) hit = hit || (s_absent[q] && s_patch[q] == id);
    if (!hit) continue;
    const unsigned char* d = a.defaults + (a.full ? (u64)p * (u64)o.v.field_bytes : 0);
    unsigned char* w = a.out + (u64)p * (u64)o.v.field_bytes;
    for (unsigned off = 0; off < o.v.field_bytes; off += 16) *reinterpret_cast<uint4*>(w + off) = *reinterpret_cast<const uint4*>(d + off);
    if (a.exists) a.exists[p] = 0;
    atomicAdd(a.stat + 2, 1u);
  }
}

#endif  // TFRA_STEP_DEVICE_PART

#ifdef TFRA_STEP_HOST_PART

struct tfra_step_driver {
  Table* t = nullptr;
  tfra_table_t* tp = nullptr;
  static constexpr unsigned NPL = 4;      // plans in rotation: batch b uses plans[b % NPL] (previous, this, next: three alive at a time)
  tfra_sparse_plan* plans[NPL] = {};
  unsigned seq = 0;                        // batches looked up so far
  bool pending = false;                    // the batch of the previous call still has to be written back
  unsigned pend_slot = 0;
  bool ahead = false;                      // plans[seq % NPL] already holds the plan of (ahead_ids, ahead_n): built by the last call
  const int64_t* ahead_ids = nullptr;
  size_t ahead_n = 0;
  SetEnt* dummy = nullptr;                 // an empty table (4 entries + the sentinel slots + padding): "no previous batch"
  unsigned* progress = nullptr;            // pinned: [0] step, [1] distinct keys the last started write-back saw
  unsigned* stat = nullptr;                // device: StepArgs::stat
  unsigned step_no = 0;
  int variant = 0;                         // TFRA_STEP_VARIANT (tuning): kernel instantiation
  unsigned long long n_overlapped = 0, n_sequential = 0;   // steps taken each way (tfra_step_driver_stats)
  unsigned why_sequential = 0;             // why the last step that was not overlapped was not (bit mask, see step_overlap_one)
};

extern "C" int tfra_step_driver_create(tfra_table_t* tp, tfra_step_driver_t** out) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !out) return set_error(TFRA_ERR_INVALID, "step_driver_create: null argument");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != t->device) { if (hipSetDevice(t->device) != hipSuccess) return set_error(TFRA_ERR_HIP, "step_driver_create: hipSetDevice"); } }
  tfra_step_driver* d = new tfra_step_driver();
  d->t = t; d->tp = tp;
  for (unsigned i = 0; i < tfra_step_driver::NPL; ++i) {
    int rc = tfra_sparse_plan_create(t->device, &d->plans[i]);
    if (rc) { tfra_step_driver_destroy(d); return rc; }
  }
  const size_t dn = 4 + 2 + SET_PAD;
  if (hipMalloc((void**)&d->dummy, dn * sizeof(SetEnt)) != hipSuccess) { d->dummy = nullptr; tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipMalloc"); }
  fill_setent_kernel<<<1, 64, 0, nullptr>>>(d->dummy, dn);
  if (hipMalloc((void**)&d->stat, 64 + 64 * 3 * 16) != hipSuccess || hipMemset(d->stat, 0, 64 + 64 * 3 * 16) != hipSuccess) { tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipMalloc"); }
  if (hipHostMalloc((void**)&d->progress, 64, hipHostMallocDefault) != hipSuccess) { d->progress = nullptr; tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipHostMalloc"); }
  d->progress[0] = d->progress[1] = 0;
  if (hipDeviceSynchronize() != hipSuccess) { tfra_step_driver_destroy(d); return set_error(TFRA_ERR_HIP, "step_driver_create: sync"); }
  const char* ev = std::getenv("TFRA_STEP_VARIANT");
  d->variant = ev ? std::atoi(ev) : 0;
  if (d->variant & 16) (void)tfra_step_driver_timing(d, nullptr);
  *out = d;
  return TFRA_OK;
}

extern "C" int tfra_step_driver_destroy(tfra_step_driver_t* d) {
  if (!d) return TFRA_OK;
  (void)hipSetDevice(d->t->device);
  (void)hipDeviceSynchronize();
  for (unsigned i = 0; i < tfra_step_driver::NPL; ++i) if (d->plans[i]) tfra_sparse_plan_destroy(d->plans[i]);
  if (d->dummy) (void)hipFree(d->dummy);
  if (d->stat) (void)hipFree(d->stat);
  if (d->progress) (void)hipHostFree(d->progress);
  delete d;
  return TFRA_OK;
}

extern "C" int tfra_step_driver_stats(const tfra_step_driver_t* d, uint64_t* overlapped, uint64_t* sequential, int* pending, uint32_t* device_counts,
                                      uint32_t* why_sequential) {
  if (d && why_sequential) *why_sequential = d->why_sequential;
  if (!d) return set_error(TFRA_ERR_INVALID, "step_driver_stats: null driver");
  if (overlapped) *overlapped = d->n_overlapped;
  if (sequential) *sequential = d->n_sequential;
  if (pending) *pending = d->pending ? 1 : 0;
  if (device_counts) {   // synchronises the device
    (void)hipSetDevice(d->t->device);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(device_counts, d->stat, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
      return set_error(TFRA_ERR_HIP, "step_driver_stats: copy");
  }
  return TFRA_OK;
}

// tuning: the role time stamps of the last <= 64 launches made with TFRA_STEP_VARIANT & 16 — out[64][3][2] = {earliest block
// start, latest block end} on the device clock (100 MHz) per launch slot (step % 64) and role (plan, write-back, lookup);
// synchronises the device and re-arms the stamps.
extern "C" int tfra_step_driver_timing(tfra_step_driver_t* d, uint64_t* out) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_driver_timing: null driver");
  (void)hipSetDevice(d->t->device);
  std::vector<uint64_t> init(64 * 3 * 2);
  for (size_t i = 0; i < init.size(); i += 2) { init[i] = ~0ULL; init[i + 1] = 0; }
  if (hipDeviceSynchronize() != hipSuccess || (out && hipMemcpy(out, d->stat + 16, init.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) ||
      hipMemcpy(d->stat + 16, init.data(), init.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
    return set_error(TFRA_ERR_HIP, "step_driver_timing: copy");
  return TFRA_OK;
}

static SetProbe probe_of(const tfra_sparse_plan* pl) { return SetProbe{pl->set_tab[pl->set_parity].ent, pl->set_m2}; }

template <bool PLAN>
static void launch_step(int variant, unsigned grid, hipStream_t s, const StepArgs& a) {
  if (variant & 16) { step_kernel<true, 2, PLAN, 1, true><<<grid, 256, 0, s>>>(a); return; }
  switch (variant & 7) {
    case 1: step_kernel<true, 1, PLAN, 1><<<grid, 256, 0, s>>>(a); break;
    case 2: step_kernel<true, 2, PLAN, 6><<<grid, 256, 0, s>>>(a); break;
    case 4: step_kernel<true, 4, PLAN, 1><<<grid, 256, 0, s>>>(a); break;
    default: step_kernel<true, 2, PLAN, 1><<<grid, 256, 0, s>>>(a); break;
  }
}
static int own_keys_per_block(int variant) { const int v = variant & 7; return v == 1 ? 16 : (v == 4 ? 64 : 32); }

// One step.  Caller holds d->t->step_mu.
static int step_overlap_one(tfra_step_driver* d, size_t n, const int64_t* ids, void* rows_out, uint8_t* exists_out, const void* defaults,
                            int default_is_full, const void* values_prev, const uint64_t* scores_prev, size_t n_next,
                            const int64_t* ids_next, hipStream_t s) {
  Table* t = d->t;
  if (n && (!ids || !rows_out || !defaults)) return set_error(TFRA_ERR_INVALID, "step_overlap: null buffer");
  if (n > MAX_IDS || n_next > MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "step_overlap: at most 2^18 ids per step");
  if (d->pending && !values_prev) return set_error(TFRA_ERR_INVALID, "step_overlap: the previous step's batch has not been written back: values_prev is null");
  if (n_next && !ids_next) return set_error(TFRA_ERR_INVALID, "step_overlap: null ids_next");
  constexpr unsigned NPL = tfra_step_driver::NPL;
  const unsigned slot = d->seq % NPL;
  tfra_sparse_plan* plan_cur = d->plans[slot];
  tfra_sparse_plan* plan_prev = d->pending ? d->plans[d->pend_slot] : nullptr;
  tfra_sparse_plan* plan_next = d->plans[(d->seq + 1) % NPL];
  std::unique_lock<std::mutex> lock(t->mu);
  int rc = t->enter(s);
  if (rc) return rc;
  // this batch's plan: built by the previous call (look-ahead), else here, in front of the step (one more launch)
  if (!(d->ahead && d->ahead_ids == ids && d->ahead_n == n)) {
    plan_cur->n = 0;
    if (n) { rc = setplan_build(plan_cur, n, ids, s, t->opts.strategy == TFRA_EVICT_LFU); if (rc) return rc; }
  }
  d->ahead = false;
  const bool aligned = (((uintptr_t)rows_out | (uintptr_t)defaults | (uintptr_t)values_prev | (size_t)t->field_bytes) & 15) == 0;
  unsigned* tags = t->ensure_own_tags(s);
  const unsigned why = (n > 0 ? 0u : 1u) | (aligned ? 0u : 2u) | (tags ? 0u : 4u) |
                       ((t->opts.aux_fields == 0 && t->opts.strategy == TFRA_EVICT_LRU && !scores_prev) ? 0u : 8u) |
                       (t->at_max_capacity() ? 0u : 16u) | (t->dense ? 0u : 32u) | ((!plan_prev || plan_prev->n > 0) ? 0u : 64u) |
                       (t->capture_safe ? 128u : 0u);
  const bool eligible = why == 0;
  if (!eligible) d->why_sequential = why;
  const unsigned step = ++d->step_no;
  if (!eligible) {
    // the same results one after the other: write-back of the previous batch, this lookup, the next batch's plan
    if (plan_prev && plan_prev->n) {
      rc = upsert_planned_impl(d->tp, plan_prev, values_prev, scores_prev, s, d->progress, step);
      if (rc) return rc;
    }
    lock.unlock();
    if (n) { rc = tfra_table_find(d->tp, n, ids, rows_out, exists_out, defaults, default_is_full, s); if (rc) return rc; }
    if (n_next) {
      rc = setplan_build(plan_next, n_next, ids_next, s, t->opts.strategy == TFRA_EVICT_LFU);
      if (rc) return rc;
    }
    d->n_sequential += 1;
  } else {
    StepArgs a{};
    OwnLaunch L{};
    const int kpb = own_keys_per_block(d->variant);
    if (plan_prev) {
      rc = own_prepare(t, plan_prev, values_prev, nullptr, s, d->progress, &L);
      if (rc) return rc;
      a.own = L.a;
      a.ctr = L.ctr; a.own_gen = L.og;
      a.own_blocks = (unsigned)std::max<size_t>(1, ((size_t)L.key_blocks * 16 + kpb - 1) / kpb);
      a.fwd = probe_of(plan_prev);
    } else {
      a.own.v = t->view_of(t->cur);
      a.own_blocks = 0;
      a.fwd = SetProbe{d->dummy, 4};
    }
    a.progress = d->progress; a.progress_val = step; a.stat = d->stat;
    a.nxt = probe_of(plan_cur);
    a.n = (unsigned)n; a.ids = (const i64*)ids; a.out = (unsigned char*)rows_out; a.exists = exists_out;
    a.defaults = (const unsigned char*)defaults; a.full = default_is_full;
    const bool fused_plan = n_next > 0 && !(d->variant & 8);
    if (n_next) {
      SetPlanLaunch P;
      if (fused_plan) {
        rc = setplan_prepare(plan_next, n_next, s, false, &P);
        if (rc) return rc;
        a.n_plan = (unsigned)n_next; a.ids_plan = (const i64*)ids_next; a.pcur = P.cur; a.pold = P.old; a.next_use_count = P.next_use_count;
        a.plan_m2 = P.m2; a.plan_blocks = P.blocks;
      } else {
        rc = setplan_build(plan_next, n_next, ids_next, s, false);   // (tuning variant: the plan as a launch of its own, in front)
        if (rc) return rc;
      }
    }
    const unsigned find_blocks = (unsigned)((n + 63) / 64);
    if (fused_plan) launch_step<true>(d->variant, a.plan_blocks + a.own_blocks + find_blocks, s, a);
    else launch_step<false>(d->variant, a.own_blocks + find_blocks, s, a);
    if (plan_prev) step_rest_kernel<<<std::min(L.rem_blocks, 512u), 256, 0, s>>>(a, reinterpret_cast<unsigned*>(L.next_ctr));   // (<= 512 blocks: all resident, see its arrival counter)
    if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "step_overlap: launch failed");
    if (plan_prev) step_epoch_public(t);
    d->n_overlapped += 1;
  }
  d->pending = n > 0;
  d->pend_slot = slot;
  if (n_next) { d->ahead = true; d->ahead_ids = ids_next; d->ahead_n = n_next; }
  d->seq += 1;
  return TFRA_OK;
}

extern "C" int tfra_table_step_overlap(tfra_step_driver_t* d, size_t n, const int64_t* ids, void* rows_out, uint8_t* exists_out,
                                       const void* defaults, int default_is_full, const void* values_prev, const uint64_t* scores_prev,
                                       size_t n_next, const int64_t* ids_next, tfra_stream_t stream) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_overlap: null driver");
  std::lock_guard<std::mutex> step_lock(d->t->step_mu);
  return step_overlap_one(d, n, ids, rows_out, exists_out, defaults, default_is_full, values_prev, scores_prev, n_next, ids_next, (hipStream_t)stream);
}

extern "C" int tfra_table_steps_overlap(tfra_step_driver_t* d, size_t count, const tfra_overlap_step* steps, tfra_stream_t stream) {
  if (!d || (count && !steps)) return set_error(TFRA_ERR_INVALID, "steps_overlap: null argument");
  std::lock_guard<std::mutex> step_lock(d->t->step_mu);
  for (size_t i = 0; i < count; ++i) {
    const tfra_overlap_step& q = steps[i];
    if (q.struct_size != sizeof(tfra_overlap_step)) return set_error(TFRA_ERR_INVALID, "steps_overlap: struct_size mismatch");
    int rc = step_overlap_one(d, q.n, q.ids, q.rows_out, q.exists_out, q.defaults, q.default_is_full, q.values_prev, q.scores_prev, q.n_next,
                              q.ids_next, (hipStream_t)stream);
    if (rc) return rc;
  }
  return TFRA_OK;
}

extern "C" int tfra_table_step_overlap_flush(tfra_step_driver_t* d, const void* values_prev, const uint64_t* scores_prev, tfra_stream_t stream) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_overlap_flush: null driver");
  std::lock_guard<std::mutex> step_lock(d->t->step_mu);
  if (!d->pending) return TFRA_OK;
  if (!values_prev) return set_error(TFRA_ERR_INVALID, "step_overlap_flush: null values_prev");
  tfra_sparse_plan* plan_prev = d->plans[d->pend_slot];
  std::lock_guard<std::mutex> lock(d->t->mu);
  int rc = upsert_planned_impl(d->tp, plan_prev, values_prev, scores_prev, stream, d->progress, ++d->step_no);
  if (rc) return rc;
  d->pending = false;
  return TFRA_OK;
}

#endif  // TFRA_STEP_HOST_PART
