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
  unsigned plan_blocks, own_blocks, find_blocks;
  int interleave;              // own blocks spread among the lookup's blocks (else: all in front of them)
  int serial_probe;            // the lookup reads the table's lines only for the ids the previous batch's plan does not hold
  unsigned* stat;              // [0] evictions the pass deferred, [1] victims the remainder noted, [2] output rows corrected
  u64* tbuf;                   // TIMING: [TIMING_SLOTS][TIMING_BLOCKS][2] block start / end stamps (nullptr otherwise)
  i64* patch_keys;             // step_rest_kernel: the launch's list of evicted keys that are ids of this batch
  unsigned* patch_count;       // its length; patch_count_next: the next step's (two alternate)
  unsigned* patch_count_next;
};

// ---- PLAN role: setplan_kernel<false> for 256 threads per 1024 ids -------------------------------------------------
// (Tried and dropped, measured inside the step on the metric's configuration: 512 ids per block with half the LDS, the tiles
// taken from the end of the batch backwards and a LOOK at the home slot before the atomics of an id that repeats in its tile
// — every block of the role runs at the same time, nobody has installed anything yet when the others look, and twice the
// blocks contend for the hot slots: the role went from 22 to 39 us and the step from 32 to 48.)
constexpr unsigned SPK_IDS = 1024, SPK_LDS = 2048, SPK_PER = SPK_LDS / 256;
struct PlanLds {
  i64 key[SPK_LDS];
  unsigned pos[SPK_LDS + 2];
  unsigned n, base;
};

// (tuning) phase stamps of the plan role: a.tbuf + TIMING_SLOTS * TIMING_BLOCKS * 2 + (launch slot * 128 + block) * 8 + k
__device__ __forceinline__ void plan_stamp(const StepArgs& a, unsigned blk, int k) {
  if (a.tbuf && threadIdx.x == 0 && blk < 128)
    a.tbuf[(size_t)64 * 4096 * 2 + ((size_t)(a.progress_val % 64u) * 128 + blk) * 8 + k] = (u64)wall_clock64();
}
__device__ __forceinline__ void plan_role(const StepArgs& a, unsigned blk, PlanLds& L) {
  const unsigned tid = threadIdx.x;
  plan_stamp(a, blk, 0);
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
  keep_live(id[0], id[1], id[2], id[3]);
  plan_stamp(a, blk, 1);   // ids arrived
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
  plan_stamp(a, blk, 2);   // LDS phase done
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
  if (a.tbuf) { keep_live(was[0], was[1], was[2], was[3]); plan_stamp(a, blk, 3); }   // first swaps back (wave 0)
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
  plan_stamp(a, blk, 4);   // every swap chain of the block resolved
  if (tid == 0) L.base = L.n ? atomicAdd(cur.count, L.n) : 0u;
  // C: empty the slots the previous build used in the OTHER table
  for (unsigned i = blk * 256u + tid; i < n_old; i += a.plan_blocks * 256u)
    *reinterpret_cast<uint4*>(old.ent + old.uslot[i]) = make_uint4(0u, 0x80000000u, 0u, 0u);
  __syncthreads();
  plan_stamp(a, blk, 5);   // list base back, other table emptied
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
  }
  if (!a.serial_probe) {   // (tuning) the table's lines travel WITH the plan probe: every id pays two random lines of the table
#pragma unroll
    for (int u = 0; u < U; ++u) {
      k0[u] = key_line(v, b0[u])[sub];
      k1[u] = key_line(v, b1[u])[sub];
    }
  }
  keep_live(e.x, e.y, e.z, e.w);
  if (!a.serial_probe) {
    keep_live(k0[0], k0[1], k0[2], k0[3]);
    keep_live(k1[0], k1[1], k1[2], k1[3]);
  }
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
  unsigned fwd_pos[U];
#pragma unroll
  for (int u = 0; u < U; ++u) fwd_pos[u] = (unsigned)__shfl((int)p1, u * 4 + grp);
  if (a.serial_probe) {
    // The table's lines BEHIND the plan probe, and only for the ids the plan does not hold: on a Zipf stream most positions of a
    // batch repeat ids of the batch before (85 % on the metric's configuration), and every line of a 273-GB table is a random,
    // TLB-missing access — the launch is bound by how many of those the memory system takes, not by wave slots.  The loads stay
    // unconditional (one wait for all of them): a forwarded id reads one hot line of the plan instead.
    const i64* hot = reinterpret_cast<const i64*>(a.fwd.ent);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      k0[u] = (fwd_pos[u] ? hot : key_line(v, b0[u]))[sub];
      k1[u] = (fwd_pos[u] ? hot : key_line(v, b1[u]))[sub];
    }
    keep_live(k0[0], k0[1], k0[2], k0[3]);
    keep_live(k1[0], k1[1], k1[2], k1[3]);
  }
  const unsigned char* src[U];
  unsigned char* dst[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned fw = fwd_pos[u];
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

// TIMING (tuning builds only): every block notes its start and end on the device clock in a.tbuf (a slot per launch and block):
// when does each role of a launch run?
constexpr unsigned TIMING_BLOCKS = 4096, TIMING_SLOTS = 64;
__device__ __forceinline__ void role_stamp(const StepArgs& a, int role, u64 t0) {
  __syncthreads();
  if (threadIdx.x == 0 && a.tbuf && blockIdx.x < TIMING_BLOCKS) {
    u64* w = a.tbuf + ((size_t)(a.progress_val % TIMING_SLOTS) * TIMING_BLOCKS + blockIdx.x) * 2;
    w[0] = t0;
    w[1] = (u64)wall_clock64();
  }
}
template <bool SIMPLE, int U, bool PLAN, bool TIMING>
__device__ __forceinline__ void step_body(const StepArgs& a) {
  unsigned b = blockIdx.x;
  const u64 t0 = TIMING ? (u64)wall_clock64() : 0;
  if constexpr (PLAN) {
    __shared__ PlanLds L;
    if (b < a.plan_blocks) { plan_role(a, b, L); if (TIMING) role_stamp(a, 0, t0); return; }
    b -= a.plan_blocks;
  }
  // The write-back's O blocks are spread evenly among the lookup's F blocks (own block j at position floor(j (O + F) / O)):
  // blocks are dispatched in index order, and a role whose blocks all come first fills every wave slot of the chip while
  // the role behind it waits for them to retire — the roles then run one after the other, not side by side.
  if (a.interleave) {
    const unsigned O = a.own_blocks, T = O + a.find_blocks;
    const unsigned c = O ? (unsigned)(((u64)b * O + T - 1) / T) : 0u;          // own blocks in front of position b
    const bool is_own = c < O && (unsigned)(((u64)c * T) / O) == b;
    if (is_own) { own_role<SIMPLE, U>(a, c); if (TIMING) role_stamp(a, 1, t0); return; }
    find_fwd_role(a, b - c);
    if (TIMING) role_stamp(a, 2, t0);
    return;
  }
  if (b < a.own_blocks) { own_role<SIMPLE, U>(a, b); if (TIMING) role_stamp(a, 1, t0); return; }
  find_fwd_role(a, b - a.own_blocks);
  if (TIMING) role_stamp(a, 2, t0);
}
// Instantiations (the SGPR budget is an attribute, not a template argument): 256-thread blocks are admitted per CU up to
// floor(800 / (ceil(sgpr / 16) * 16 + 16)) — 106 scalar registers (what the big argument block costs) allow 6, 96 allow 7, 80 allow 8.
#define TFRA_STEP_KERNEL(NAME, UU, PLAN, TIMING, NSGPR)                                                                  \
  __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(NSGPR))) void NAME(const StepArgs a) { step_body<true, UU, PLAN, TIMING>(a); }
TFRA_STEP_KERNEL(step_k_u2, 2, true, false, 104)
TFRA_STEP_KERNEL(step_k_u1, 1, true, false, 104)
TFRA_STEP_KERNEL(step_k_u4, 4, true, false, 104)
TFRA_STEP_KERNEL(step_k_u1_s96, 1, true, false, 96)
TFRA_STEP_KERNEL(step_k_u1_s80, 1, true, false, 80)
TFRA_STEP_KERNEL(step_k_u2_t, 2, true, true, 104)
TFRA_STEP_KERNEL(step_k_u1_t, 1, true, true, 104)
TFRA_STEP_KERNEL(step_k_u1_s80_t, 1, true, true, 80)
TFRA_STEP_KERNEL(step_k_u2_np, 2, false, false, 104)
TFRA_STEP_KERNEL(step_k_u1_np, 1, false, false, 104)
TFRA_STEP_KERNEL(step_k_u1_s80_np, 1, false, false, 80)
#undef TFRA_STEP_KERNEL

// ---- the remainder of a step: left-over keys of the pass + corrections of the lookup's output ------------------------
// upsert_rest_kernel<16, SRC_SET> over the pass's item list (the flags of all keys when the list overflowed).  Runs AFTER the
// launch that held the lookup, so an eviction here may hit a key that lookup has just returned a row for: every victim that is
// one of this batch's ids (plan `nxt`) goes onto a list of the launch.  When every block has finished its items (one arrival
// counter: the grid is at most 512 blocks, all resident, and every block arrives without waiting for anything) and the list is
// not empty, ALL blocks look the listed victims up again and, for those that are absent now, rewrite the lookup's output rows
// with the default row and clear their exists flags — what a lookup behind the write-back returns — each block for its share
// of the batch's positions.  (A victim that is present again was a key of the previous batch whose own left-over write came
// later in this kernel.)  The first form let the one block that had noted a victim scan the whole batch: 170 us for 131 072 ids,
// and right after a bulk load in rank order the hottest ids are the least recently used entries: one step in eight paid it.
constexpr unsigned PATCH_CAP = 64, PATCH_GCAP = 4096;
__global__ __launch_bounds__(256) void step_rest_kernel(const StepArgs a, unsigned* zero4) {
  __shared__ i64 s_patch[PATCH_CAP];
  __shared__ unsigned char s_absent[PATCH_CAP];
  __shared__ unsigned s_nv;
  const OwnArgs& o = a.own;
  const unsigned* slow_ctr = &a.ctr->n_a;
  unsigned* arrived = &a.ctr->spare[0];
  const unsigned gi = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const OwnItem* it0 = o.items + (gi < o.item_cap ? gi : 0u);
  const uint4 f0 = reinterpret_cast<const uint4*>(it0)[0], f1 = reinterpret_cast<const uint4*>(it0)[1];
  const unsigned counted = *slow_ctr;
  if (zero4 && blockIdx.x == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0;   // last kernel of this use: arm the next use's counters
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.patch_count_next = 0;            // (the next step's list: its last reader is long gone)
  if (counted == 0) return;                                                    // no items, no evictions, nothing to correct
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
        const unsigned at = atomicAdd(a.patch_count, 1u);
        atomicAdd(a.stat + 1, 1u);
        if (at < PATCH_GCAP) __hip_atomic_store(a.patch_keys + at, vk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else atomicAdd(o.v.err_count, 1u);   // (thousands of such victims in one step: reported by check_errors, never silent)
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) { fresh += __shfl_xor(fresh, off); failed += __shfl_xor(failed, off); }
  if (lane == 0) {
    if (fresh) size_add(o.v, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, fresh);
    if (failed) atomicAdd(o.v.err_count, (unsigned)failed);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's table stores and list entries (write-through / agent-scope) are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = false;
    for (unsigned it = 0; it < (1u << 22) && !ok; ++it) {
      ok = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x;
      if (!ok) __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) atomicAdd(o.v.err_count, 1u);   // (never seen: every block arrives without waiting for anything)
    s_nv = min(__hip_atomic_load(a.patch_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), PATCH_GCAP);
  }
  __syncthreads();
  const unsigned nv = s_nv;
  for (unsigned base = 0; base < nv; base += PATCH_CAP) {
    const unsigned np = min(PATCH_CAP, nv - base);
    // which of these victims are absent now?  (16 groups, one victim each per round; coherent loads)
    for (unsigned q = threadIdx.x >> 4; q < np; q += 16) {
      const i64 vk = __hip_atomic_load(a.patch_keys + base + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const i64 row = probe_find<true>(o.v, vk, sub, gshift);
      if (sub == 0) { s_patch[q] = vk; s_absent[q] = row < 0 ? 1 : 0; }
    }
    __syncthreads();
    // those, against this block's share of the batch
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < a.n; p += gridDim.x * blockDim.x) {
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
    __syncthreads();
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
  u64* tbuf = nullptr;                     // device: StepArgs::tbuf (TFRA_STEP_VARIANT & 16)
  unsigned last_rest_step = ~0u;           // step number of the last step_rest_kernel launch (it zeroes the next step's victim counter)
  unsigned char* patch = nullptr;          // device: two counters (one 128-B line each) + two lists of PATCH_GCAP keys (step_rest_kernel)
  unsigned tinfo[64][3] = {};              // per launch slot: plan blocks, own blocks, grid
  unsigned step_no = 0;
  int variant = 0;                         // TFRA_STEP_VARIANT (tuning): kernel instantiation
  unsigned long long n_overlapped = 0, n_sequential = 0;   // steps taken each way (tfra_step_driver_stats)
  unsigned why_sequential = 0;             // why the last step that was not overlapped was not (bit mask, see step_overlap_one)
  std::vector<hipEvent_t> kev;             // tfra_step_driver_time_kernels: 3 events per timed step (before / between / behind its two launches)
  size_t kev_left = 0, kev_used = 0;
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
  if (hipMalloc((void**)&d->patch, 256 + 2 * PATCH_GCAP * 8) != hipSuccess || hipMemset(d->patch, 0, 256 + 2 * PATCH_GCAP * 8) != hipSuccess) { d->patch = nullptr; tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipMalloc"); }
  if (hipMalloc((void**)&d->stat, 64) != hipSuccess || hipMemset(d->stat, 0, 64) != hipSuccess) { tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipMalloc"); }
  if (hipHostMalloc((void**)&d->progress, 64, hipHostMallocDefault) != hipSuccess) { d->progress = nullptr; tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipHostMalloc"); }
  d->progress[0] = d->progress[1] = 0;
  if (hipDeviceSynchronize() != hipSuccess) { tfra_step_driver_destroy(d); return set_error(TFRA_ERR_HIP, "step_driver_create: sync"); }
  const char* ev = std::getenv("TFRA_STEP_VARIANT");
  d->variant = ev ? std::atoi(ev) : 0;
  if (d->variant & 16) {
    const size_t bytes = (size_t)TIMING_SLOTS * TIMING_BLOCKS * 16 + (size_t)64 * 128 * 8 * 8;
    if (hipMalloc((void**)&d->tbuf, bytes) != hipSuccess || hipMemset(d->tbuf, 0, bytes) != hipSuccess) { d->tbuf = nullptr; tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipMalloc"); }
  }
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
  if (d->tbuf) (void)hipFree(d->tbuf);
  if (d->patch) (void)hipFree(d->patch);
  for (hipEvent_t e : d->kev) (void)hipEventDestroy(e);
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

// tuning: the block time stamps of the last <= 64 launches made with TFRA_STEP_VARIANT & 16, reduced per role —
// out[64][3][2] = {earliest block start, latest block end} on the device clock (100 MHz) per launch slot (step % 64) and role
// (plan, write-back, lookup), ~0 / 0 where nothing ran; synchronises the device and re-arms the stamps.
extern "C" int tfra_step_driver_timing(tfra_step_driver_t* d, uint64_t* out) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_driver_timing: null driver");
  if (!d->tbuf) return set_error(TFRA_ERR_INVALID, "step_driver_timing: the driver was not created with TFRA_STEP_VARIANT & 16");
  (void)hipSetDevice(d->t->device);
  const size_t words = (size_t)TIMING_SLOTS * TIMING_BLOCKS * 2;
  std::vector<uint64_t> h(words);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), d->tbuf, words * 8, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemset(d->tbuf, 0, words * 8) != hipSuccess)
    return set_error(TFRA_ERR_HIP, "step_driver_timing: copy");
  {   // plan phase stamps: median over launches and blocks of (stamp k - stamp 0), in ticks, behind the role spans
    std::vector<uint64_t> ph((size_t)64 * 128 * 8);
    const size_t off = (size_t)TIMING_SLOTS * TIMING_BLOCKS * 2;
    if (hipMemcpy(ph.data(), d->tbuf + off, ph.size() * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemset(d->tbuf + off, 0, ph.size() * 8) != hipSuccess)
      return set_error(TFRA_ERR_HIP, "step_driver_timing: copy");
    if (out) {
      for (int k = 1; k <= 5; ++k) {
        std::vector<uint64_t> dlt;
        for (size_t i = 0; i < (size_t)64 * 128; ++i) if (ph[i * 8] && ph[i * 8 + k]) dlt.push_back(ph[i * 8 + k] - ph[i * 8]);
        std::sort(dlt.begin(), dlt.end());
        out[64 * 3 * 2 + (k - 1)] = dlt.empty() ? 0 : dlt[dlt.size() / 2];
      }
    }
  }
  if (!out) return TFRA_OK;
  for (unsigned sl = 0; sl < TIMING_SLOTS; ++sl) {
    const unsigned pb = d->tinfo[sl][0], ob = d->tinfo[sl][1], grid = std::min(d->tinfo[sl][2], TIMING_BLOCKS);
    for (int r = 0; r < 3; ++r) { out[(sl * 3 + r) * 2] = ~0ULL; out[(sl * 3 + r) * 2 + 1] = 0; }
    for (unsigned b = 0; b < grid; ++b) {
      const uint64_t t0 = h[((size_t)sl * TIMING_BLOCKS + b) * 2], t1 = h[((size_t)sl * TIMING_BLOCKS + b) * 2 + 1];
      if (!t1) continue;
      int r = b < pb ? 0 : (b < pb + ob ? 1 : 2);
      if (b >= pb && !(d->variant & 32)) {   // own blocks spread among the lookup's (step_body)
        const unsigned bb = b - pb, O = ob, T = d->tinfo[sl][2] - pb;
        const unsigned c = O ? (unsigned)(((uint64_t)bb * O + T - 1) / T) : 0u;
        r = (c < O && (unsigned)(((uint64_t)c * T) / O) == bb) ? 1 : 2;
      }
      out[(sl * 3 + r) * 2] = std::min(out[(sl * 3 + r) * 2], t0);
      out[(sl * 3 + r) * 2 + 1] = std::max(out[(sl * 3 + r) * 2 + 1], t1);
    }
    d->tinfo[sl][2] = 0;
  }
  return TFRA_OK;
}

// Measurement: HIP events around the two launches of the next `steps` overlapped steps (on the stream they are launched on);
// tfra_step_driver_kernel_times then waits for them and returns the average duration of each launch in microseconds.
extern "C" int tfra_step_driver_time_kernels(tfra_step_driver_t* d, size_t steps) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_driver_time_kernels: null driver");
  (void)hipSetDevice(d->t->device);
  while (d->kev.size() < steps * 3) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return set_error(TFRA_ERR_HIP, "step_driver_time_kernels: event");
    d->kev.push_back(e);
  }
  d->kev_left = steps; d->kev_used = 0;
  return TFRA_OK;
}
extern "C" int tfra_step_driver_kernel_times(tfra_step_driver_t* d, double* step_kernel_us, double* rest_kernel_us, size_t* steps) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_driver_kernel_times: null driver");
  double a = 0, b = 0;
  for (size_t i = 0; i < d->kev_used; ++i) {
    float x = 0, y = 0;
    if (hipEventSynchronize(d->kev[i * 3 + 2]) != hipSuccess || hipEventElapsedTime(&x, d->kev[i * 3], d->kev[i * 3 + 1]) != hipSuccess ||
        hipEventElapsedTime(&y, d->kev[i * 3 + 1], d->kev[i * 3 + 2]) != hipSuccess)
      return set_error(TFRA_ERR_HIP, "step_driver_kernel_times: event");
    a += x; b += y;
  }
  const double nn = d->kev_used ? (double)d->kev_used : 1.0;
  if (step_kernel_us) *step_kernel_us = a / nn * 1e3;
  if (rest_kernel_us) *rest_kernel_us = b / nn * 1e3;
  if (steps) *steps = d->kev_used;
  d->kev_left = 0; d->kev_used = 0;
  return TFRA_OK;
}

static SetProbe probe_of(const tfra_sparse_plan* pl) { return SetProbe{pl->set_tab[pl->set_parity].ent, pl->set_m2}; }

// variant (TFRA_STEP_VARIANT, tuning): bits 0-2 kernel (0 U=2 | 1 U=1 | 3 U=1, 80 SGPRs | 4 U=4 | 5 U=1, 96 SGPRs),
// 8 the plan as a launch of its own, 16 time stamps
static void launch_step(int variant, bool plan, unsigned grid, hipStream_t s, const StepArgs& a) {
  const int k = variant & 7;
  if (!plan) {
    if (k == 1 || k == 5) step_k_u1_np<<<grid, 256, 0, s>>>(a);
    else if (k == 3) step_k_u1_s80_np<<<grid, 256, 0, s>>>(a);
    else step_k_u2_np<<<grid, 256, 0, s>>>(a);
  } else if (variant & 16) {
    if (k == 1 || k == 5) step_k_u1_t<<<grid, 256, 0, s>>>(a);
    else if (k == 3) step_k_u1_s80_t<<<grid, 256, 0, s>>>(a);
    else step_k_u2_t<<<grid, 256, 0, s>>>(a);
  } else {
    switch (k) {
      case 1: step_k_u1<<<grid, 256, 0, s>>>(a); break;
      case 3: step_k_u1_s80<<<grid, 256, 0, s>>>(a); break;
      case 4: step_k_u4<<<grid, 256, 0, s>>>(a); break;
      case 5: step_k_u1_s96<<<grid, 256, 0, s>>>(a); break;
      default: step_k_u2<<<grid, 256, 0, s>>>(a); break;
    }
  }
}
static int own_keys_per_block(int variant) { const int v = variant & 7; return (v == 1 || v == 3 || v == 5) ? 16 : (v == 4 ? 64 : 32); }

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
    a.progress = d->progress; a.progress_val = step; a.stat = d->stat; a.tbuf = d->tbuf;
    a.patch_count = reinterpret_cast<unsigned*>(d->patch + 128 * (step & 1u)); a.patch_count_next = reinterpret_cast<unsigned*>(d->patch + 128 * ((step & 1u) ^ 1u));
    a.patch_keys = reinterpret_cast<i64*>(d->patch + 256) + (size_t)PATCH_GCAP * (step & 1u);
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
        a.plan_m2 = P.m2; a.plan_blocks = (unsigned)((n_next + SPK_IDS - 1) / SPK_IDS);   // (= P.blocks: 1024 ids per block)
      } else {
        rc = setplan_build(plan_next, n_next, ids_next, s, false);   // (tuning variant: the plan as a launch of its own, in front)
        if (rc) return rc;
      }
    }
    const unsigned find_blocks = (unsigned)((n + 63) / 64);
    a.find_blocks = find_blocks; a.interleave = (d->variant & 32) ? 0 : 1; a.serial_probe = (d->variant & 64) ? 0 : 1;
    if (d->tbuf) { unsigned* ti = d->tinfo[step % TIMING_SLOTS]; ti[0] = fused_plan ? a.plan_blocks : 0; ti[1] = a.own_blocks; ti[2] = ti[0] + ti[1] + find_blocks; }
    const bool timed = d->kev_left > 0 && plan_prev;
    if (timed) (void)hipEventRecord(d->kev[d->kev_used * 3], s);
    launch_step(d->variant, fused_plan, (fused_plan ? a.plan_blocks : 0u) + a.own_blocks + find_blocks, s, a);
    if (timed) (void)hipEventRecord(d->kev[d->kev_used * 3 + 1], s);
    if (plan_prev && d->last_rest_step + 1 != step && hipMemsetAsync(a.patch_count, 0, 4, s) != hipSuccess) return set_error(TFRA_ERR_HIP, "step_overlap: memset");
    if (plan_prev) d->last_rest_step = step;
    if (plan_prev) step_rest_kernel<<<std::max(std::min(L.rem_blocks, 512u), 128u), 256, 0, s>>>(a, reinterpret_cast<unsigned*>(L.next_ctr));   // (128 .. 512 blocks: all resident, see its arrival counter; the corrections take the whole grid)
    if (timed) { (void)hipEventRecord(d->kev[d->kev_used * 3 + 2], s); d->kev_used += 1; d->kev_left -= 1; }
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
