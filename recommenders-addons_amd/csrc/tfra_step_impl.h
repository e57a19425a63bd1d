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
// ONE launch per step, six roles by block index (block-uniform branches; grid order = dispatch order):
//   BUILD    one 2048-slot window of plan(i+2)'s table per block, in LDS, from the segments a SCATTER filled one launch ago
//   SCATTER  one tile of 1024 ids of batch i+3 per block: distinct (id, last position) pairs into per-(window, tile) segments
//            (a plan is built over two launches WITHOUT a global atomic; ids two batches ahead)
//   MAP      (round 5) 1024 positions of the NEXT lookup per block, probed in this batch's plan and sorted by where their row will
//            come from (forwarded / table): the next launch's lookup reads its entries instead of probing, whole waves of one kind
//   OWN      write-back(i): own_batch16 over slices of plan(i)'s table (~40 keys per block), victims checked against plan(i+1)
//   FIND     lookup(i+1) with forwarding from plan(i) / values_i, driven by the list the previous launch's MAP made
//   TAIL     the last 32 blocks: wait for the OWN blocks (a counter), take the keys the pass left over (lost claims, deferred
//            evictions) with the locked protocol, note every victim that batch i+1 looks up, and — if any — wait for the FIND
//            blocks and correct their output rows.  Nothing a TAIL block waits for waits for anything itself.
// One stream, no events, no host in the loop: the sequence is enqueued many steps ahead (tfra_table_steps_overlap).
// Measurements on the metric's configuration (10^9 slots, 131 072 Zipf-1.2 ids, 22.5 K distinct): round 4 31-33 us per launch
// (roles ALONE, TFRA_STEP_ABLATE: builders 9.6 us, lookup + builders 19.8, write-back + tail 23.0); round 5 21.3-21.5 us on average
// under rocprofv3 — the MAP role (31.0 -> 26.0 us on one box) and 448-slot write-back slices (26.2 -> 20.8): 12 K waves that wait on
// memory 75 % of their time, 92.4 MB of HBM traffic for 80.0 MB algorithmic; DESIGN.md sections 4.5 and 5.

#ifdef TFRA_STEP_DEVICE_PART

struct StepArgs {
  OwnArgs own;                 // write-back of the PREVIOUS batch (its keys come from `fwd`, the previous batch's plan); own_blocks == 0: none pending
  OwnCtrs* ctr;                // the plan's left-over counters (the step counts in sync[1]; these belong to the sequential path)
  unsigned own_gen;
  unsigned* progress;          // pinned: [0] step
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
  // BUILD role: the plan of the NEXT batch, one window of its table per block, from the segments a scatter filled one launch ago
  SetEnt* build_ent;           // the table (every slot is written)
  unsigned build_m2, build_tiles, build_blocks;
  const SetEnt* build_pairs; const unsigned* build_cnt; const SetEnt* build_ovf; const unsigned* build_ovf_cnt;
  unsigned* build_ucnt;        // [windows] distinct keys per window of the table being built (plain stores: no atomic on a hot word —
                               // 586 fire-and-forget adds from the write-back's blocks to ONE counter cost the launch 4 us)
  const unsigned* own_ucnt;    // the same array of the plan being written back (nullptr: not built by a step launch); block 0 sums it
  unsigned own_ucnt_n;         // into progress[1] (pinned): the host sizes later launches' slices from it
  // SCATTER role: the batch after next, one tile of 1024 ids per block -> per (window, tile) segments
  unsigned scat_n, scat_m2, scat_tiles, scat_blocks;
  const i64* scat_ids;
  SetEnt* scat_pairs; unsigned* scat_cnt; SetEnt* scat_ovf; unsigned* scat_ovf_cnt; unsigned* scat_ovf_cnt_next;   // (two overflow counters alternate: this scatter zeroes the next one's)
  // MAP role: the positions of the NEXT batch, sorted by where their row will come from — probed in `nxt` (THIS batch's plan, the
  // one the next lookup forwards from): entries {position, last position + 1 in this batch | 0, key}; per segment of 1024 positions
  // the table entries from the front, the forwarded ones from the back.  find_list != nullptr: the list a MAP role made of THIS batch one launch ago
  unsigned map_n, map_blocks;
  const i64* map_ids;
  uint4* map_out;
  const uint4* find_list;
  unsigned own_slice;          // plan slots per write-back block
  unsigned find_first;         // lookup blocks dispatched in FRONT of the write-back's (the list's table-bound chunks: the long chains start with the launch)
  unsigned own_blocks, find_blocks, tail_blocks;
  unsigned* sync;              // this launch's counters, one 128-byte line each: [0] write-back blocks done, [32 .. 32*8] lookup blocks done
                               // (8 shards), [32*9] tail blocks through their items; sync_next: the next launch's (two sets alternate).  [1] the
                               // write-back's left-over keys (item list length), [32*9+1] = patch_count: each read WITH the arrivals beside it
  unsigned* sync_next;
  unsigned* zero4;             // the left-over counters of the plan's NEXT use (armed by the tail)
  int ablate;                  // (tuning) roles that return at once: 1 builders, 2 write-back + tail, 4 lookup, 8 tail
  int serial_probe;            // the lookup reads the table's lines only for the ids the previous batch's plan does not hold
  unsigned* stat;              // [0] evictions the pass deferred, [1] victims the remainder noted, [2] output rows corrected
  u64* tbuf;                   // TIMING: [TIMING_SLOTS][TIMING_BLOCKS][2] block start / end stamps (nullptr otherwise)
  i64* patch_keys;             // tail: the launch's list of keys whose slot changed hands and which this batch looks up
  unsigned* patch_count;       // its length; patch_count_next: the next step's (two alternate)
  unsigned* patch_count_next;
};

// LDS of a block, whatever its role (24.6 KB): a 2048-slot table (scatter: the tile's ids; build: the window) or, for the
// write-back, the compacted keys of its slice of the plan
struct StepLds {
  i64 key[SET_WIN];
  unsigned pos[SET_WIN + 2];   // (+2: the two sentinel key values)
  unsigned cnt[256];           // scatter: pairs per window
  unsigned n;
};
// Plan slots per write-back block (StepArgs::own_slice; TFRA_STEP_OWN_SLICE overrides it for tuning, <= 768).  Rounds 3-4 used 288
// (~25 keys at 22.7 K keys in 2^18 slots: one round of the block's four waves; 911 blocks) — the write-back's blocks, dispatched in
// front of the lookup's, then took every wave slot of the chip for the first 6 us of the launch.  With the lookup's blocks short
// (MAP lists, below) the balance is elsewhere: 448 slots = 586 blocks of ~39 keys (two rounds) leave 300 slots to the lookup from
// the first microsecond, and the write-back itself ENDS EARLIER (16 us instead of 19: fewer waves contend for the same lines).
// Measured on the metric's configuration, one box: 288 -> 26.2 us per step, 384 -> 21.3, 448 -> 21.1, 512 -> 20.8 / 22.4 (two boxes),
// 576 -> 24.1, 768 -> 29.2.  The optimum follows the keys per block, not the slots: the driver sizes the slice for ~40 keys from the
// distinct-key count of an earlier batch (the tail leaves it in pinned memory), between 96 and 512 slots — a batch of all-distinct
// ids (m2 = 2 n: every second slot taken) gets 96-slot slices instead of 224 keys = seven rounds per block; configs[2]'s 78 K keys
// per batch 128 (measured there, per-step with 65 K evictions: 96 -> 72 us, 160 -> 76, 288 -> 66, 448 -> 88).
constexpr unsigned OWN_SLICE_DEFAULT = 448;

// ---- SCATTER role: setplan_kernel's LDS phase, then plain stores ------------------------------------------------------
// Equal ids of the tile meet in LDS (compare-and-swap on the key, max on position + 1); every distinct id then goes, with its
// last position in the tile, into the segment (window of its home slot, this tile): a position inside the segment from an LDS
// counter, a plain 16-byte store.  No global atomic (an id-heavy window — 32 distinct ids of ONE tile in ONE of the >= 128
// windows — spills into an overflow list with one).  The round-3 plan kernel put every distinct id of every tile into ONE global
// table with a compare-and-swap and a max: 100 K device-scope atomics per batch, the hot ids' slots taking one from every tile
// — inside the step's launch each such round trip took 7 us (queues at the hot slots) and the role 21 us, as long as lookup
// and write-back together.
__device__ __forceinline__ void scatter_role(const StepArgs& a, unsigned tile, StepLds& L) {
  const unsigned tid = threadIdx.x;
  const unsigned wins = a.scat_m2 >> SET_WIN_LOG2;
  for (unsigned i = tid; i < SET_WIN + 2; i += 256) { if (i < SET_WIN) L.key[i] = EMPTY_KEY; L.pos[i] = 0; }
  L.cnt[tid] = 0;
  if (tile == 0 && tid == 0) *a.scat_ovf_cnt_next = 0;
  i64 id[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned g = tile * 1024u + (unsigned)r * 256u + tid;
    id[r] = g < a.scat_n ? a.scat_ids[g] : 0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned g = tile * 1024u + (unsigned)r * 256u + tid;
    if (g >= a.scat_n) continue;
    unsigned slot;
    if (is_reserved_key(id[r])) slot = SET_WIN + (unsigned)reserved_index(id[r]);
    else {
      slot = (unsigned)(fmix64((u64)id[r]) >> 41) & (SET_WIN - 1);
      for (;;) {
        const i64 was = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&L.key[slot]), (unsigned long long)EMPTY_KEY, (unsigned long long)id[r]);
        if (was == EMPTY_KEY || was == id[r]) break;
        slot = (slot + 1) & (SET_WIN - 1);
      }
    }
    atomicMax(&L.pos[slot], g + 1u);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < (int)(SET_WIN / 256); ++r) {
    const unsigned sl = tid + (unsigned)r * 256u;
    const unsigned p1 = L.pos[sl];
    if (!p1) continue;
    const i64 key = L.key[sl];
    const unsigned w = ((unsigned)(fmix64((u64)key) >> 20) & (a.scat_m2 - 1)) >> SET_WIN_LOG2;
    const unsigned at = atomicAdd(&L.cnt[w], 1u);
    const uint4 pair = make_uint4((unsigned)(u64)key, (unsigned)((u64)key >> 32), p1, 0u);
    if (at < SEG_CAP) *reinterpret_cast<uint4*>(a.scat_pairs + ((size_t)w * a.scat_tiles + tile) * SEG_CAP + at) = pair;
    else *reinterpret_cast<uint4*>(a.scat_ovf + atomicAdd(a.scat_ovf_cnt, 1u)) = pair;
  }
  if (tid < 2 && L.pos[SET_WIN + tid] != 0) {   // a sentinel key value occurred in this tile: window 0 carries it
    const i64 key = EMPTY_KEY + (i64)tid;
    const unsigned at = atomicAdd(&L.cnt[0], 1u);
    const uint4 pair = make_uint4((unsigned)(u64)key, (unsigned)((u64)key >> 32), L.pos[SET_WIN + tid], 0u);
    if (at < SEG_CAP) *reinterpret_cast<uint4*>(a.scat_pairs + ((size_t)0 * a.scat_tiles + tile) * SEG_CAP + at) = pair;
    else *reinterpret_cast<uint4*>(a.scat_ovf + atomicAdd(a.scat_ovf_cnt, 1u)) = pair;
  }
  __syncthreads();
  if (tid < wins) a.scat_cnt[(size_t)tid * a.scat_tiles + tile] = min(L.cnt[tid], SEG_CAP);
}

// ---- BUILD role: one window of the next batch's plan ---------------------------------------------------------------------
// Thread t takes the pairs tile t left for this window (a handful), inserts them into the window's image in LDS (linear probing
// inside the window, max on the position: the rule every prober follows, set_at) and the block writes the image out — all
// 2048 slots, so the table needs no emptying between its uses.  Two dependent round trips (counts, pairs), no atomics.
__device__ __forceinline__ void build_role(const StepArgs& a, unsigned win, StepLds& L) {
  const unsigned tid = threadIdx.x;
  for (unsigned i = tid; i < SET_WIN + 2; i += 256) { if (i < SET_WIN) L.key[i] = EMPTY_KEY; L.pos[i] = 0; }
  if (tid == 0) L.n = 0;
  const unsigned c = tid < a.build_tiles ? a.build_cnt[(size_t)win * a.build_tiles + tid] : 0u;
  const unsigned novf = *a.build_ovf_cnt;
  __syncthreads();
  auto insert = [&](i64 key, unsigned p1) {
    unsigned slot;
    if (is_reserved_key(key)) slot = SET_WIN + (unsigned)reserved_index(key);
    else {
      slot = ((unsigned)(fmix64((u64)key) >> 20) & (a.build_m2 - 1)) & (SET_WIN - 1);
      for (;;) {
        const i64 was = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&L.key[slot]), (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (was == EMPTY_KEY || was == key) break;
        slot = (slot + 1) & (SET_WIN - 1);
      }
    }
    atomicMax(&L.pos[slot], p1);
  };
  const SetEnt* seg = a.build_pairs + ((size_t)win * a.build_tiles + tid) * SEG_CAP;
  for (unsigned j = 0; j < c; j += 4) {   // (c <= SEG_CAP; four pairs in flight)
    uint4 pr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pr[q] = *reinterpret_cast<const uint4*>(seg + min(j + (unsigned)q, c - 1));
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (j + (unsigned)q < c) insert((i64)(((u64)pr[q].y << 32) | pr[q].x), pr[q].z);
  }
  for (unsigned i = tid; i < novf; i += 256) {   // (an adversarial batch)
    const uint4 pr = *reinterpret_cast<const uint4*>(a.build_ovf + i);
    const i64 key = (i64)(((u64)pr.y << 32) | pr.x);
    const unsigned w = is_reserved_key(key) ? 0u : ((unsigned)(fmix64((u64)key) >> 20) & (a.build_m2 - 1)) >> SET_WIN_LOG2;
    if (w == win) insert(key, pr.z);
  }
  __syncthreads();
  SetEnt* out = a.build_ent + (size_t)win * SET_WIN;
  const unsigned wslots = a.build_m2 < SET_WIN ? a.build_m2 : SET_WIN;
  unsigned occ = 0;
  for (unsigned sl = tid; sl < wslots; sl += 256) {
    const i64 k = L.key[sl];
    const unsigned p1 = L.pos[sl];
    occ += p1 != 0;
    *reinterpret_cast<uint4*>(out + sl) = make_uint4((unsigned)(u64)k, (unsigned)((u64)k >> 32), p1, 0u);
  }
  if (a.build_ucnt) {   // distinct keys of this window (for the host's choice of the write-back's slice size, two launches from now)
    for (int off = 32; off > 0; off >>= 1) occ += (unsigned)__shfl_xor((int)occ, off);
    if ((tid & 63) == 0) atomicAdd(&L.n, occ);
    __syncthreads();
    if (tid == 0) a.build_ucnt[win] = L.n;
  }
  if (win == 0 && tid < 2) {   // the two sentinel slots behind the table: key word = "taken" marker
    const unsigned p1 = L.pos[SET_WIN + tid];
    *reinterpret_cast<uint4*>(a.build_ent + a.build_m2 + tid) = p1 ? make_uint4(1u, 0u, p1, 0u) : make_uint4(0u, 0x80000000u, 0u, 0u);
  }
}

// ---- FIND role: find_kernel<16, 4, WT, PF1> + forwarding -----------------------------------------------------------
// A wave takes KW = 4 U ids (U = 4: 16, U = 8: 32): lane j < KW — and its 64 / KW replicas — holds id j and hashes it; for the
// plan probe the replicas of an id read consecutive entries of its chain (one 16-B load per lane, all 64 lanes busy, no
// redundancy); the table probe is find_kernel's, one 16-lane group per id, U ids per group in flight.
// (Tried, U = 8: 32 ids per wave, twice the loads in flight in registers the write-back's roles need anyway — a lookup block then
// takes 14.6 us instead of 7.4 for twice the ids, the step 34.5 us instead of 31.9: what a wave waits for is served at a rate, not
// after a latency.  SQ counters of the launch: 12.8 K waves, 300 vector + 204 scalar instructions per wave, 77 % of the wave cycles
// parked on s_waitcnt, L2 hit ratio 24 % — plan, value rows and table lines of eight XCDs' worth of batch do not fit eight 4-MB L2s.)
template <int N, typename T>
__device__ __forceinline__ void keep_live_n(T (&x)[N]) {
  static_assert(N % 4 == 0, "keep_live_n: multiples of four");
#pragma unroll
  for (int q = 0; q < N; q += 4) keep_live(x[q], x[q + 1], x[q + 2], x[q + 3]);
}
template <int U>
__device__ __forceinline__ void find_fwd_role(const StepArgs& a, unsigned blk) {
  constexpr int KW = 4 * U, REP = 64 / KW;
  const TableView& v = a.own.v;
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, grp = lane >> 4;
  const int rep = lane / KW;                       // which of the id's replicas this lane is
  const unsigned wave = blk * 4u + (threadIdx.x >> 6);
  const unsigned base = wave * (unsigned)KW;
  if (base >= a.n) return;
  const unsigned last = a.n - 1;
  const i64 kreg = a.ids[min(base + (unsigned)(lane & (KW - 1)), last)];
  u64 hreg;
  const unsigned b0reg = (unsigned)bucket0(kreg, v.nb, hreg);
  const unsigned b1reg = (unsigned)bucket1(hreg, b0reg, v.nb);
  const bool resv = is_reserved_key(kreg);
  const unsigned home = set_home(a.fwd, kreg, hreg);
  const unsigned wm = set_wmask(a.fwd.m2);
  const unsigned eidx = resv ? home + (unsigned)rep : set_at(home, (unsigned)rep, wm);
  uint4 e = *reinterpret_cast<const uint4*>(a.fwd.ent + eidx);
  i64 key[U], k0[U], k1[U];
  unsigned b0[U], b1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int j = u * 4 + grp;
    key[u] = shfl_i64(kreg, j);
    b0[u] = (unsigned)__shfl((int)b0reg, j);
    b1[u] = (unsigned)__shfl((int)b1reg, j);
  }
  keep_live(e.x, e.y, e.z, e.w);
  // the plan probe, per id: a match in any of its replicas' entries is the id; none and no EMPTY among them: go on (rare)
  const i64 ekey = (i64)(((u64)e.y << 32) | e.x);
  const bool match = resv ? (rep == 0 && ekey != EMPTY_KEY) : ekey == kreg;
  unsigned p1 = match ? e.z : 0u;
  unsigned stop = (ekey == EMPTY_KEY || resv) ? 1u : 0u;
#pragma unroll
  for (int o = KW; o < 64; o <<= 1) { p1 |= (unsigned)__shfl_xor((int)p1, o); stop |= (unsigned)__shfl_xor((int)stop, o); }
  if (!p1 && !stop) {
    for (unsigned t = REP; t <= wm; ++t) {
      const SetEnt* q = a.fwd.ent + set_at(home, t, wm);
      const i64 k = q->key;
      if (k == kreg) { p1 = q->pos1; break; }
      if (k == EMPTY_KEY) break;
    }
  }
  unsigned fwd_pos[U];
#pragma unroll
  for (int u = 0; u < U; ++u) fwd_pos[u] = (unsigned)__shfl((int)p1, u * 4 + grp);
  {
    // The table's lines BEHIND the plan probe, and only for the ids the plan does not hold: on a Zipf stream most positions of a
    // batch repeat ids of the batch before (85 % on the metric's configuration), and every line of a 273-GB table is a random,
    // TLB-missing access.  The loads stay unconditional (one wait for all of them): a forwarded id reads one hot line of the plan.
    // (Tried: the lines of every id in flight WITH the plan probe — one trip less, two random lines more per forwarded id: no gain.)
    const i64* hot = reinterpret_cast<const i64*>(a.fwd.ent);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      k0[u] = (fwd_pos[u] ? hot : key_line(v, b0[u]))[sub];
      k1[u] = (fwd_pos[u] ? hot : key_line(v, b1[u]))[sub];
    }
    keep_live_n<U>(k0);
    keep_live_n<U>(k1);
  }
  const unsigned char* src[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned fw = fwd_pos[u];
    const unsigned idx = min(base + (unsigned)(u * 4 + grp), last);
    i64 word = 0;
    if (!fw) word = probe_find_word(v, key[u], b0[u], b1[u], k0[u], sub, gshift, &k1[u]);
    // (write-through like the rows: the tail's corrections come from another workgroup and must land BEHIND this store)
    if (a.exists && sub == 0) __hip_atomic_store(a.exists + idx, (uint8_t)(fw != 0 || word >= 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    src[u] = fw ? a.own.vals + (u64)(fw - 1u) * (u64)v.field_bytes
                : (word >= 0 ? word_row_ptr(v, (u64)word) : a.defaults + (a.full ? (u64)idx * (u64)v.field_bytes : 0));
  }
  for (unsigned off = sub * 16; off < v.field_bytes; off += 256) {
    uint4 tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) tmp[u] = *reinterpret_cast<const uint4*>(src[u] + off);
    keep_live_n<U>(tmp);
#pragma unroll
    for (int u = 0; u < U; ++u)
      store_wt16(a.out + (u64)min(base + (unsigned)(u * 4 + grp), last) * (u64)v.field_bytes + off, tmp[u]);
  }
}
// ---- MAP role (round 5): where will each position of the NEXT batch get its row from? -----------------------------------------
// The lookup's chain was ids -> plan entry -> table lines -> rows -> stores, and 85 % of a Zipf batch's positions stop at the plan
// entry (forwarded from the rows being written) — but a wave of 16 ids nearly always holds one that does not, so every wave went
// the whole way.  The plan the next lookup forwards from (this batch's) is complete when this launch starts and the next ids are
// announced: one launch AHEAD, off the lookup's chain, a block probes 1024 positions in it and writes them out as 16-byte entries
// {position, last position + 1 | 0, key} into ITS segment of the list — positions that need the table packed from the front of
// the segment, forwarded ones from its back: ranks by ballot and LDS, no atomic, no counter.  The next launch's lookup reads its
// entries with its first (coalesced) trip; all but one wave per segment are of one kind, and a wave of forwarded positions goes
// entry -> value rows -> stores: no hash, no plan probe, no table line.
// (First form: ONE list, packed from both ends with two returned atomics per block — 31.0 -> 26.8 us per step on one box, its MAP
// blocks 10 us long, most of it the atomics' round trip.)
constexpr unsigned MAP_SEG = 1024;
__device__ __forceinline__ void map_role(const StepArgs& a, unsigned tile, StepLds& L) {
  const unsigned tid = threadIdx.x;
  const int lane = tid & 63;
  const unsigned wave = tid >> 6;
  i64 id[4];
  unsigned p1[4];
  bool valid[4];
  const unsigned wm = set_wmask(a.nxt.m2);
  unsigned home[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned g = tile * MAP_SEG + (unsigned)r * 256u + tid;
    valid[r] = g < a.map_n;
    id[r] = a.map_ids[valid[r] ? g : a.map_n - 1];
  }
  uint4 e[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    home[r] = set_home(a.nxt, id[r], fmix64((u64)id[r]));
    e[r] = *reinterpret_cast<const uint4*>(a.nxt.ent + home[r]);
  }
  keep_live(e[0], e[1], e[2], e[3]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const i64 ek = (i64)(((u64)e[r].y << 32) | e[r].x);
    if (is_reserved_key(id[r])) p1[r] = ek != EMPTY_KEY ? e[r].z : 0u;
    else if (ek == id[r]) p1[r] = e[r].z;
    else if (ek == EMPTY_KEY) p1[r] = 0u;
    else {   // (rare: the plan's table is < 10 % full) further down the chain, inside the window
      p1[r] = 0u;
      for (unsigned t = 1; t <= wm; ++t) {
        const SetEnt* q = a.nxt.ent + set_at(home[r], t, wm);
        const i64 k = q->key;
        if (k == id[r]) { p1[r] = q->pos1; break; }
        if (k == EMPTY_KEY) break;
      }
    }
  }
  // ranks inside the segment: per round and wave by ballot, across the waves by LDS
  unsigned rank[4], nf_w = 0, nt_w = 0;
  const u64 below = (1ULL << lane) - 1ULL;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const u64 mf = __ballot(valid[r] && p1[r] != 0), mt = __ballot(valid[r] && p1[r] == 0);
    rank[r] = p1[r] ? nf_w + (unsigned)__popcll(mf & below) : nt_w + (unsigned)__popcll(mt & below);
    nf_w += (unsigned)__popcll(mf); nt_w += (unsigned)__popcll(mt);
  }
  if (lane == 0) { L.cnt[wave] = nf_w; L.cnt[4 + wave] = nt_w; }
  __syncthreads();
  unsigned bf = 0, bt = 0;
  for (unsigned w = 0; w < wave; ++w) { bf += L.cnt[w]; bt += L.cnt[4 + w]; }
  const unsigned seg0 = tile * MAP_SEG, m = min(MAP_SEG, a.map_n - seg0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (!valid[r]) continue;
    const unsigned g = seg0 + (unsigned)r * 256u + tid;
    const unsigned at = p1[r] ? seg0 + m - 1u - (bf + rank[r]) : seg0 + bt + rank[r];
    a.map_out[at] = make_uint4(g, p1[r], (unsigned)(u64)id[r], (unsigned)((u64)id[r] >> 32));
  }
}

// ---- FIND role from a MAP list: T tiles of 16 entries per wave --------------------------------------------------------------
// Block b takes chunk b / nseg of segment b % nseg (chunk-major: the FRONT chunks of every segment — the table-bound positions, the
// long chains — are the first blocks of the role; the all-forwarded chunks fill in behind them).  A wave takes T consecutive tiles of
// its chunk, the next tile's entries loaded before the current tile's rows are stored.
template <int U, int T>
__device__ __forceinline__ void find_list_role(const StepArgs& a, unsigned blk) {
  constexpr int KW = 4 * U;
  constexpr unsigned CH = 64u * (unsigned)T;   // entries per block
  const TableView& v = a.own.v;
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, grp = lane >> 4;
  const unsigned nseg = (a.n + MAP_SEG - 1) / MAP_SEG;
  const unsigned sg = blk % nseg, ch = blk / nseg;
  const unsigned base0 = sg * MAP_SEG + ch * CH + (threadIdx.x >> 6) * (unsigned)(KW * T);
  if (base0 >= a.n) return;
  const unsigned last = a.n - 1;
  uint4 ent = a.find_list[min(base0 + (unsigned)(lane & (KW - 1)), last)];
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    const unsigned base = base0 + (unsigned)(t * KW);
    if (base >= a.n) break;
    const uint4 cur = ent;
    if (t + 1 < T) ent = a.find_list[min(base + (unsigned)KW + (unsigned)(lane & (KW - 1)), last)];
    const bool all_fwd = __ballot(cur.y == 0) == 0ULL;
    unsigned pos[U], fwd_pos[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = u * 4 + grp;
      pos[u] = (unsigned)__shfl((int)cur.x, j);
      fwd_pos[u] = (unsigned)__shfl((int)cur.y, j);
    }
    const unsigned char* src[U];
    if (all_fwd) {   // (wave-uniform) every row of this tile is one of the rows being written
#pragma unroll
      for (int u = 0; u < U; ++u) {
        src[u] = a.own.vals + (u64)(fwd_pos[u] - 1u) * (u64)v.field_bytes;
        if (a.exists && sub == 0) __hip_atomic_store(a.exists + pos[u], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      const i64 kreg = (i64)(((u64)cur.w << 32) | cur.z);
      u64 hreg;
      const unsigned b0reg = (unsigned)bucket0(kreg, v.nb, hreg);
      const unsigned b1reg = (unsigned)bucket1(hreg, b0reg, v.nb);
      i64 key[U], k0[U], k1[U];
      unsigned b0[U], b1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = u * 4 + grp;
        key[u] = shfl_i64(kreg, j);
        b0[u] = (unsigned)__shfl((int)b0reg, j);
        b1[u] = (unsigned)__shfl((int)b1reg, j);
      }
      const i64* hot = reinterpret_cast<const i64*>(a.fwd.ent);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        k0[u] = (fwd_pos[u] ? hot : key_line(v, b0[u]))[sub];
        k1[u] = (fwd_pos[u] ? hot : key_line(v, b1[u]))[sub];
      }
      keep_live_n<U>(k0);
      keep_live_n<U>(k1);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned fw = fwd_pos[u];
        i64 word = 0;
        if (!fw) word = probe_find_word(v, key[u], b0[u], b1[u], k0[u], sub, gshift, &k1[u]);
        if (a.exists && sub == 0) __hip_atomic_store(a.exists + pos[u], (uint8_t)(fw != 0 || word >= 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        src[u] = fw ? a.own.vals + (u64)(fw - 1u) * (u64)v.field_bytes
                    : (word >= 0 ? word_row_ptr(v, (u64)word) : a.defaults + (a.full ? (u64)pos[u] * (u64)v.field_bytes : 0));
      }
    }
    for (unsigned off = sub * 16; off < v.field_bytes; off += 256) {
      uint4 tmp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) tmp[u] = *reinterpret_cast<const uint4*>(src[u] + off);
      keep_live_n<U>(tmp);
#pragma unroll
      for (int u = 0; u < U; ++u) store_wt16(a.out + (u64)pos[u] * (u64)v.field_bytes + off, tmp[u]);
    }
  }
}
// a lookup block is done: its output rows are in memory (write-through, acknowledged).  Only the tail's rare corrections wait for this.
__device__ __forceinline__ void find_arrive(const StepArgs& a) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && a.tail_blocks) __hip_atomic_fetch_add(a.sync + 32 * (1 + (blockIdx.x & 7u)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- OWN role: the ownership pass over the previous batch's plan, read straight from its TABLE ---------------------------
// The plan keeps no dense list of its keys any more (that list was a returned atomic per tile and a dependent round trip of
// the builder): the block takes a slice of own_slice slots of the plan's table (one 16-byte load per thread and a few more),
// compacts the occupied ones in LDS — key, last position, slot (the key's flag byte) — and its four waves take them 4 U at a
// time through own_batch16, victims checked against this batch's plan.
template <bool SIMPLE, int U>
__device__ __forceinline__ void own_role(const StepArgs& a, unsigned blk, StepLds& L) {
  const OwnArgs& o = a.own;
  const unsigned tid = threadIdx.x;
  const int lane = tid & 63;
  const unsigned lo = blk * a.own_slice, total_slots = a.fwd.m2 + 2;
  if (tid == 0) L.n = 0;
  uint4 e[3];
  bool have[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const unsigned off = tid + (unsigned)r * 256u;
    have[r] = off < a.own_slice && lo + off < total_slots;
    e[r] = *reinterpret_cast<const uint4*>(a.fwd.ent + (have[r] ? lo + off : lo));
  }
  if (blk == 0 && tid == 0 && a.progress) __hip_atomic_store(a.progress, a.progress_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const bool publish = blk == 0 && a.own_ucnt && a.progress;   // (block-uniform)
  unsigned ucnt = 0;
  if (publish) {
    if (tid == 0) L.cnt[200] = 0;
    ucnt = tid < a.own_ucnt_n ? a.own_ucnt[tid] : 0u;
  }
  __syncthreads();
  if (publish) {
    for (int off = 32; off > 0; off >>= 1) ucnt += (unsigned)__shfl_xor((int)ucnt, off);
    if (lane == 0) atomicAdd(&L.cnt[200], ucnt);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const unsigned slot = lo + tid + (unsigned)r * 256u;
    const i64 k = (i64)(((u64)e[r].y << 32) | e[r].x);
    if (have[r] && k != EMPTY_KEY) {
      const unsigned at = atomicAdd(&L.n, 1u);
      L.key[at] = slot >= a.fwd.m2 ? EMPTY_KEY + (i64)(slot - a.fwd.m2) : k;   // (sentinel slots hold a marker, the slot says which key)
      L.pos[at] = e[r].z;
      L.pos[1024 + at] = slot;
    }
  }
  __syncthreads();
  const unsigned cnt = L.n;
  if (publish && tid == 0) __hip_atomic_store(a.progress + 1, L.cnt[200], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // this batch's distinct keys
  // (Tried: ONE round per block — 16 U keys — and the few keys beyond it straight onto the item list, so that no block runs the
  // chain of round trips twice: the slowest write-back blocks are not the ones with a second round — the role ended at 19 us as
  // before — and the tail, with 60 items instead of a dozen, ran longer: 31.9 -> 37.3 us per step.)
  int fresh = 0;
  const OwnFlags fl = own_setup<SIMPLE>(o);
  for (unsigned wbase = (tid >> 6) * (4 * U); wbase < cnt; wbase += 4 * (4 * U)) {
    const unsigned i = wbase + (unsigned)(lane & 15);
    const unsigned c = min(i, cnt - 1);
    own_batch16<16, SIMPLE, SRC_GIVEN, U, true, false, true>(o, fl, L.pos[1024 + c], (lane & 15) < 4 * U && i < cnt, a.own_gen, a.sync + 1, lane, fresh, &a.nxt,
                                                             a.stat, L.key[c], L.pos[c] - 1u, &a.fwd);
  }
  for (int off = 32; off > 0; off >>= 1) fresh += __shfl_xor(fresh, off);
  if (lane == 0 && fresh) size_add(o.v, blk * 4u + (tid >> 6), fresh);
  // this block's part of the pass is in memory (everything it wrote went write-through or by atomics): the tail may go on
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (Tried: the write-back over a DENSE KEY LIST of the plan — a plan built by setplan_kernel as a launch of its own — instead of
// slices of its table: 8 keys for every wave, no empty slots to skip — 43.5 us per step against 45.0 with that plan launch, 33 without.)

// TIMING (tuning builds only): every block notes its start and end on the device clock in a.tbuf (a slot per launch and block):
// when does each role of a launch run?
constexpr unsigned TIMING_BLOCKS = 4096, TIMING_SLOTS = 64;
__device__ __forceinline__ void role_stamp(const StepArgs& a, u64 t0) {
  __syncthreads();
  if (threadIdx.x == 0 && a.tbuf && blockIdx.x < TIMING_BLOCKS) {
    u64* w = a.tbuf + ((size_t)(a.progress_val % TIMING_SLOTS) * TIMING_BLOCKS + blockIdx.x) * 2;
    w[0] = t0;
    w[1] = (u64)wall_clock64();
  }
}
// role of block b: 0 build, 1 scatter, 2 write-back, 3 lookup, 4 tail; *idx = its index in the role.  Grid order: the builders
// (short chains of dependent round trips, few blocks), the whole write-back (the tail waits for it), the lookup, the TAIL.
// Tried, measured on the metric's configuration, dropped: (a) the write-back's blocks spread evenly among the lookup's — no
// change; (b) the tail IN FRONT of the lookup's blocks, so that it holds its wave slots from the start instead of getting them
// 18 us into the launch: its 32 polling blocks (one lane each, s_sleep between polls) slow the write-back they wait for from 19
// to 34 us — the step went from 33 to 45 us.  Behind the lookup's blocks the tail starts when the write-back is (nearly) done
// and hardly ever polls; (c) the write-back's blocks in front of the builders': no change.
__host__ __device__ __forceinline__ int step_role(unsigned b, unsigned build_blocks, unsigned scat_blocks, unsigned map_blocks, unsigned F1, unsigned O,
                                                  unsigned F, unsigned T, unsigned* idx) {
  (void)T;
  if (b < build_blocks) { *idx = b; return 0; }
  b -= build_blocks;
  if (b < scat_blocks) { *idx = b; return 1; }
  b -= scat_blocks;
  if (b < map_blocks) { *idx = b; return 5; }   // (MAP: numbered behind the roles the tuning scripts know)
  b -= map_blocks;
  if (b < F1) { *idx = b; return 3; }            // the lookup's first blocks in front of the write-back's
  b -= F1;
  if (b < O) { *idx = b; return 2; }
  b -= O;
  if (b < F - F1) { *idx = F1 + b; return 3; }
  *idx = b - (F - F1);
  return 4;
}
__device__ __forceinline__ void tail_role(const StepArgs& a, unsigned blk, StepLds& L);
// A kernel gets the registers of its hungriest role for EVERY wave: the lookup and the builders need 58, the write-back 91, the
// tail's locked protocol 113 — 4 blocks per CU for 32 blocks of the grid.  The kernels are compiled for 5 waves per SIMD
// (amdgpu_waves_per_eu: 96 registers, six spilled ones, all in the tail): 1280 resident blocks instead of 1024, the lookup's
// first blocks start with the launch instead of behind the builders — 33.3 -> 31.4 us per step.  6 waves (80 registers, 62
// spilled, some in the write-back): 32.6 us.
template <bool SIMPLE, int U, bool TIMING, int WAVES>
__device__ __forceinline__ void step_body(const StepArgs& a) {
  __shared__ StepLds L;
  const u64 t0 = TIMING ? (u64)wall_clock64() : 0;
  unsigned idx;
  const int role = step_role(blockIdx.x, a.build_blocks, a.scat_blocks, a.map_blocks, a.find_first, a.own_blocks, a.find_blocks, a.tail_blocks, &idx);
  if (role == 0) { if (!(a.ablate & 1)) build_role(a, idx, L); }
  else if (role == 1) { if (!(a.ablate & 1)) scatter_role(a, idx, L); }
  else if (role == 5) { if (!(a.ablate & 1)) map_role(a, idx, L); }
  else if (role == 2) { if (!(a.ablate & 2)) own_role<SIMPLE, U>(a, idx, L); }
  else if (role == 3) { if (!(a.ablate & 4)) { if (a.find_list) find_list_role<4, 1>(a, idx); else find_fwd_role<4>(a, idx); } find_arrive(a); }
  else { if (!(a.ablate & (2 | 8))) tail_role(a, idx, L); }
  if (TIMING) role_stamp(a, t0);
}
// Instantiations (the budgets are attributes, not template arguments): 256-thread blocks are admitted per CU up to
// floor(800 / (ceil(sgpr / 16) * 16 + 16)) — ~106 scalar registers (what the big argument block costs) allow 6, 96 allow 7, 80 allow 8.
#define TFRA_STEP_KERNEL(NAME, SIMPLE, UU, TIMING, W)                                                                \
  __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(104), amdgpu_waves_per_eu(W, W))) void NAME(const StepArgs a) { step_body<SIMPLE, UU, TIMING, W>(a); }
TFRA_STEP_KERNEL(step_k_u2, true, 2, false, 5)
TFRA_STEP_KERNEL(step_k_u2_t, true, 2, true, 5)
TFRA_STEP_KERNEL(step_k_u1, true, 1, false, 5)        // (tuning) 4 keys per wave in the write-back
TFRA_STEP_KERNEL(step_k_u2_w4, true, 2, false, 4)     // (tuning) no spills, 4 blocks per CU
TFRA_STEP_KERNEL(step_k_u2_w6, true, 2, false, 6)     // (tuning) 6 blocks per CU
// Round 6: the strategy read at run time (SIMPLE = false: own_setup takes it from the table's ScoreP) — EPOCHLRU tables, whose scores
// are (epoch << 32 | clock): like LRU a new key is ALWAYS admitted (it carries the highest score of its buckets), which is what the
// forwarding rests on; LFU / EPOCHLFU / CUSTOMIZED may refuse a key and stay on the sequential path.  4 blocks per CU (no spills).
TFRA_STEP_KERNEL(step_k_gen, false, 2, false, 4)
#undef TFRA_STEP_KERNEL

// ---- TAIL role: the remainder of a step INSIDE its launch ------------------------------------------------------------------
// The keys the pass left over (lost claims, evictions it deferred) with the locked protocol, and the corrections of the
// lookup's output.  As a launch of its own behind the step this cost 5.5 us of dependent round trips plus two kernel
// boundaries — a quarter of the step — for a dozen keys.  Here it is the LAST blocks of the grid (blocks are dispatched in index
// order: when a tail block runs, every other block of the launch is running or done — nothing it waits for can be waiting for a
// wave slot), which
//   1. wait until every write-back block has arrived (a.sync[0]; their stores are write-through, acknowledged before they arrive),
//   2. take the item list with the locked protocol, beside the lookup blocks still running — an eviction here may hit a key the
//      lookup has returned (or is about to return) a row for: every victim that is one of this batch's ids (plan `nxt`) goes
//      onto the launch's list;
//   3. meet (a.sync[32*9]); if the list is not empty — rare — wait for every lookup block too, look the listed victims up again
//      and, for those that are absent now, rewrite the lookup's output rows with the default row and clear their exists flags,
//      each block for its share of the batch.  (A victim that is present again was a key of the previous batch whose own
//      left-over write came later.)
// The spins are bounded; a timeout is reported through the table's error counter, never silent.
constexpr unsigned PATCH_CAP = 64, PATCH_GCAP = 4096, TAIL_BLOCKS = 32;
__device__ __forceinline__ bool spin_until(const unsigned* ctr, unsigned want) {
  for (unsigned it = 0; it < (1u << 22); ++it) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    __builtin_amdgcn_s_sleep(8);
  }
  return false;
}
// the same on the LOW word of an 8-byte pair, read as one: *both = the pair as it was when the low word had reached `want` (a count
// kept in the high word by those who arrive is complete then — their additions to it come before their arrival)
__device__ __forceinline__ bool spin_until_pair(const unsigned* ctr, unsigned want, unsigned long long* both) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(ctr);
  for (unsigned it = 0; it < (1u << 22); ++it) {
    const unsigned long long v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)v >= want) { *both = v; return true; }
    __builtin_amdgcn_s_sleep(8);
  }
  *both = 0;
  return false;
}
__device__ __forceinline__ uint4 load_coherent16(const void* p) {   // written write-through by another workgroup of this launch
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint4((unsigned)x, (unsigned)(x >> 32), (unsigned)y, (unsigned)(y >> 32));
}
__device__ __forceinline__ void tail_role(const StepArgs& a, unsigned blk, StepLds& L) {
  i64* s_patch = L.key;                                            // [PATCH_CAP]
  i64* s_row = L.key + PATCH_CAP;                                  // [PATCH_CAP] row index now, -1: absent
  unsigned& s_nv = L.n;
  const OwnArgs& o = a.own;
  const unsigned tid = threadIdx.x;
  const int lane = tid & 63, sub = lane & 15, gshift = lane & 48;
  if (blk == 0) {   // arm what the next launch counts in (its users of two launches ago are long gone)
    if (tid < 4 && a.zero4) a.zero4[tid] = 0;
    if (tid >= 32 && tid < 42) a.sync_next[32 * (tid - 32)] = 0;
    if (tid == 62) a.sync_next[1] = 0;          // (the left-over count beside the write-back's arrivals)
    if (tid == 63) *a.patch_count_next = 0;     // (the victim count beside the tail's arrivals)
  }
  bool ok = true;
  // (the chain from here on is the end of the step — the lookup is done by the time the items are: the left-over count arrives WITH
  // the write-back's last arrival, the victim count with the tail's.  Measured: no change, 8 dependent trips are left; neither did
  // an item's value row fetched along with its bucket lines)
  if (tid == 0) {
    unsigned long long both;
    ok = spin_until_pair(a.sync, a.own_blocks, &both);
    L.cnt[1] = (unsigned)(both >> 32);
  }
  if (tid == 0) { L.cnt[2] = ok ? 1u : 0u; if (!ok) atomicAdd(o.v.err_count, 1u); }   // a timeout leaves keys unwritten: reported, never silent
  __syncthreads();
  const unsigned counted = L.cnt[1];
  ok = L.cnt[2] != 0;
  if (counted == 0) return;                                        // no items, no evictions, nothing to correct (the same in every tail block)
  const bool listed = counted <= o.item_cap;
  const unsigned n = listed ? counted : a.fwd.m2 + 2;              // (the list overflowed: the flag byte of every slot of the plan's table)
  const unsigned gi = (blk * 256u + tid) >> 4, ngroups = (a.tail_blocks * 256u) >> 4;
  int fresh = 0, failed = 0;
  // a key whose slot changed hands — or showed LOCKED for a moment — while the lookup was running, and which the lookup looks for
  auto note = [&](i64 k) {
    if (k == EMPTY_KEY || k == LOCKED_KEY || !a.n || !set_contains_group(a.nxt, k, sub, gshift)) return;
    if (sub == 0) {
      const unsigned at = atomicAdd(a.patch_count, 1u);
      atomicAdd(a.stat + 1, 1u);
      if (at < PATCH_GCAP) __hip_atomic_store(a.patch_keys + at, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else atomicAdd(o.v.err_count, 1u);   // (thousands of them in one step: reported by check_errors, never silent)
    }
  };
  for (unsigned i = gi; i < n; i += ngroups) {
    i64 vk = EMPTY_KEY, gb[4];
    int ngb = 0;
    if (listed) {
      const uint4 w0 = load_coherent16(o.items + i), w1 = load_coherent16(reinterpret_cast<const unsigned char*>(o.items + i) + 16);
      const i64 key = (i64)(((u64)w0.y << 32) | w0.x);
      locked_upsert_kv<16>(o.v, o.vals, key, w0.z, ((u64)w1.y << 32) | w1.x, o.ai, o.sp, sub, gshift, fresh, failed, w1.z != 0, w1.w, &vk, 0, 0, gb, &ngb);
      if (sub == 0) o.dflag[w0.w] = 0;
    } else {
      if (__hip_atomic_load(o.dflag + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 4) continue;
      const SetEnt* pe = a.fwd.ent + i;
      const i64 key = i >= a.fwd.m2 ? EMPTY_KEY + (i64)(i - a.fwd.m2) : pe->key;
      locked_upsert_kv<16>(o.v, o.vals, key, pe->pos1 - 1, 1, o.ai, o.sp, sub, gshift, fresh, failed, false, 0, &vk, 0, 0, gb, &ngb);
      if (sub == 0) o.dflag[i] = 0;
    }
    note(vk);
    for (int q = 0; q < ngb && q < 4; ++q) note(gb[q]);
    if (ngb > 4 && sub == 0) atomicAdd(o.v.err_count, 1u);
  }
  for (int off = 32; off > 0; off >>= 1) { fresh += __shfl_xor(fresh, off); failed += __shfl_xor(failed, off); }
  if (lane == 0) {
    if (fresh) size_add(o.v, (blk * 256u + tid) >> 6, fresh);
    if (failed) atomicAdd(o.v.err_count, (unsigned)failed);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's table stores and list entries (write-through / agent-scope) are acknowledged
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(a.sync + 32 * 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long both;
    ok = spin_until_pair(a.sync + 32 * 9, a.tail_blocks, &both) && ok;
    unsigned nv = min((unsigned)(both >> 32), PATCH_GCAP);   // a.patch_count, the word beside the arrivals
    if (nv) {   // the corrections overwrite what the lookup wrote: every lookup block must be done
      unsigned have = 0;
      for (unsigned it = 0; it < (1u << 20) && have < a.find_blocks; ++it) {
        have = 0;
        for (int q = 1; q <= 8; ++q) have += __hip_atomic_load(a.sync + 32 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (have < a.find_blocks) __builtin_amdgcn_s_sleep(16);
      }
      ok = ok && have >= a.find_blocks;
    }
    if (!ok) atomicAdd(o.v.err_count, 1u);   // (never seen: nothing anybody waits for here waits for anything itself)
    s_nv = nv;
  }
  __syncthreads();
  const unsigned nv = s_nv;
  for (unsigned base = 0; base < nv; base += PATCH_CAP) {
    const unsigned np = min(PATCH_CAP, nv - base);
    // where are these keys now?  (16 groups, one key each per round; coherent loads)  The lookup is done again for them:
    for (unsigned q = tid >> 4; q < np; q += 16) {
      const i64 vk = __hip_atomic_load(a.patch_keys + base + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const i64 row = probe_find<true>(o.v, vk, sub, gshift);
      if (sub == 0) { s_patch[q] = vk; s_row[q] = row; }
    }
    __syncthreads();
    // ... at this block's share of the batch's positions: the row the table holds now, or the default row
    for (unsigned p = blk * 256u + tid; p < a.n; p += a.tail_blocks * 256u) {
      const i64 id = a.ids[p];
      int hit = -1;
      for (unsigned q = 0; q < np; ++q) if (s_patch[q] == id) hit = (int)q;
      if (hit < 0) continue;
      const i64 row = s_row[hit];
      const unsigned char* d = row >= 0 ? row_ptr(o.v, row) : a.defaults + (a.full ? (u64)p * (u64)o.v.field_bytes : 0);
      unsigned char* w = a.out + (u64)p * (u64)o.v.field_bytes;
      for (unsigned off = 0; off < o.v.field_bytes; off += 16) *reinterpret_cast<uint4*>(w + off) = load_coherent16(d + off);   // (a row another workgroup of this launch may have written)
      if (a.exists) a.exists[p] = row >= 0;
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
  static constexpr unsigned NPL = 4;      // plans in rotation: batch b uses plans[b % NPL] (previous, this, next, the one being scattered)
  tfra_sparse_plan* plans[NPL] = {};
  unsigned seq = 0;                        // batches looked up so far
  bool pending = false;                    // the batch of the previous call still has to be written back
  unsigned pend_slot = 0;
  const int64_t* pend_ids = nullptr;       // its ids (the caller keeps them until the batch has been written back)
  size_t pend_n = 0;
  bool ahead = false;                      // plans[seq % NPL] already holds the plan of (ahead_ids, ahead_n): built by the last call
  const int64_t* ahead_ids = nullptr;
  size_t ahead_n = 0;
  unsigned scat_uses = 0;                  // scatters so far (the overflow counter of a plan's segments alternates)
  SetEnt* dummy = nullptr;                 // an empty table (4 entries + the sentinel slots + padding): "no previous batch"
  unsigned* progress = nullptr;            // pinned: [0] step
  unsigned* stat = nullptr;                // device: StepArgs::stat
  u64* tbuf = nullptr;                     // device: StepArgs::tbuf (TFRA_STEP_VARIANT & 16)
  unsigned tinfo[64][7] = {};              // per launch slot: build, scatter, own, lookup blocks, grid (the rest: tail), map blocks, lookup blocks in front of the write-back
  int find_first = 0;                      // TFRA_STEP_FIND_FIRST (tuning): lookup blocks in front of the write-back's
  unsigned own_slice = OWN_SLICE_DEFAULT;  // TFRA_STEP_OWN_SLICE (tuning): plan slots per write-back block (<= 768)
  bool own_slice_fixed = false;            // ... given: no adaptation to the batch's distinct-key count
  // MAP lists (round 5): two buffers of up to MAX_IDS 16-byte entries alternate — one is read by this launch's lookup, the other filled for the next
  unsigned char* mapbuf = nullptr;         // device: 2 x cap entries
  size_t map_cap = 0;
  unsigned map_slot = 0;                   // the list filled last
  bool map_valid = false;                  // ... and what it holds: the positions of (map_ids, map_n) probed in map_plan's current table
  const int64_t* map_ids = nullptr;
  size_t map_n = 0;
  const tfra_sparse_plan* map_plan = nullptr;
  unsigned map_gen = 0;                    // (the plan's build generation at that time)
  unsigned long long n_find_listed = 0;    // lookups served from a MAP list
  unsigned last_tail_step = ~0u;           // step number of the last launch that had a tail (it zeroes the next launch's counters)
  unsigned char* patch = nullptr;          // device: two victim counters (one 128-B line each) + two lists of PATCH_GCAP keys + two sets of 10 sync counters (tail_role)
  unsigned step_no = 0;
  int ablate = 0; unsigned ablate_after = 40;   // TFRA_STEP_ABLATE / TFRA_STEP_ABLATE_AFTER (tuning: timing of the roles alone; results are wrong)
  int variant = 0;                         // TFRA_STEP_VARIANT (tuning): kernel instantiation
  unsigned long long n_overlapped = 0, n_sequential = 0;   // steps taken each way (tfra_step_driver_stats)
  unsigned long long n_built_in_launch = 0, n_built_in_front = 0;   // plans of the next batch built by the step launch / by a launch of their own
  unsigned why_sequential = 0;             // why the last step that was not overlapped was not (bit mask, see step_overlap_one)
  std::vector<hipEvent_t> kev;             // tfra_step_driver_time_kernels: 3 events per timed step (before / behind its launch; the third marks the end of the step)
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
  constexpr size_t PATCH_BYTES = 256 + 2 * PATCH_GCAP * 8 + 2 * 10 * 128;
  if (hipMalloc((void**)&d->patch, PATCH_BYTES) != hipSuccess || hipMemset(d->patch, 0, PATCH_BYTES) != hipSuccess) { d->patch = nullptr; tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipMalloc"); }
  if (hipMalloc((void**)&d->stat, 64) != hipSuccess || hipMemset(d->stat, 0, 64) != hipSuccess) { tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipMalloc"); }
  if (hipHostMalloc((void**)&d->progress, 64, hipHostMallocDefault) != hipSuccess) { d->progress = nullptr; tfra_step_driver_destroy(d); return set_error(TFRA_ERR_OOM, "step_driver_create: hipHostMalloc"); }
  d->progress[0] = d->progress[1] = 0;
  if (hipDeviceSynchronize() != hipSuccess) { tfra_step_driver_destroy(d); return set_error(TFRA_ERR_HIP, "step_driver_create: sync"); }
  const char* ev = std::getenv("TFRA_STEP_VARIANT");
  d->variant = ev ? std::atoi(ev) : 0;
  if (const char* ab = std::getenv("TFRA_STEP_ABLATE")) d->ablate = std::atoi(ab);
  if (const char* ab = std::getenv("TFRA_STEP_ABLATE_AFTER")) d->ablate_after = (unsigned)std::atoi(ab);
  if (const char* os_ = std::getenv("TFRA_STEP_OWN_SLICE")) { d->own_slice = std::min(768u, std::max(64u, (unsigned)std::atoi(os_))); d->own_slice_fixed = true; }
  if (const char* ff = std::getenv("TFRA_STEP_FIND_FIRST")) d->find_first = std::atoi(ff);
  if (d->variant & 16) {
    const size_t bytes = (size_t)TIMING_SLOTS * TIMING_BLOCKS * 16;
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
  if (d->mapbuf) (void)hipFree(d->mapbuf);
  for (hipEvent_t e : d->kev) (void)hipEventDestroy(e);
  if (d->progress) (void)hipHostFree(d->progress);
  delete d;
  return TFRA_OK;
}

extern "C" int tfra_step_driver_stats(const tfra_step_driver_t* d, uint64_t* overlapped, uint64_t* sequential, int* pending, uint32_t* device_counts,
                                      uint32_t* why_sequential, uint64_t* plans_built) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_driver_stats: null driver");
  if (why_sequential) *why_sequential = d->why_sequential;
  if (overlapped) *overlapped = d->n_overlapped;
  if (sequential) *sequential = d->n_sequential;
  if (pending) *pending = d->pending ? 1 : 0;
  if (plans_built) { plans_built[0] = d->n_built_in_launch; plans_built[1] = d->n_built_in_front; }
  if (device_counts) {   // synchronises the device
    (void)hipSetDevice(d->t->device);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(device_counts, d->stat, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
      return set_error(TFRA_ERR_HIP, "step_driver_stats: copy");
  }
  return TFRA_OK;
}

extern "C" int tfra_step_driver_lookups_listed(const tfra_step_driver_t* d, uint64_t* out) {
  if (!d || !out) return set_error(TFRA_ERR_INVALID, "step_driver_lookups_listed: null argument");
  *out = d->n_find_listed;
  return TFRA_OK;
}

// tuning: the block time stamps of the last <= 64 launches made with TFRA_STEP_VARIANT & 16, reduced per role —
// out[64][6][4] = {earliest block start, latest block end, median block duration, 95th percentile of the block durations} on the
// device clock (100 MHz) per launch slot (step % 64) and role (build, scatter, write-back, lookup, tail, map), ~0 / 0 where nothing ran;
// synchronises the device and re-arms the stamps.
extern "C" int tfra_step_driver_timing(tfra_step_driver_t* d, uint64_t* out) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_driver_timing: null driver");
  if (!d->tbuf) return set_error(TFRA_ERR_INVALID, "step_driver_timing: the driver was not created with TFRA_STEP_VARIANT & 16");
  (void)hipSetDevice(d->t->device);
  const size_t words = (size_t)TIMING_SLOTS * TIMING_BLOCKS * 2;
  std::vector<uint64_t> h(words);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), d->tbuf, words * 8, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemset(d->tbuf, 0, words * 8) != hipSuccess)
    return set_error(TFRA_ERR_HIP, "step_driver_timing: copy");
  if (!out) return TFRA_OK;
  for (unsigned sl = 0; sl < TIMING_SLOTS; ++sl) {
    const unsigned* ti = d->tinfo[sl];
    const unsigned grid = std::min(ti[4], TIMING_BLOCKS);
    constexpr int NR = 6;
    auto role_of = [&](unsigned b) { unsigned idx; return step_role(b, ti[0], ti[1], ti[5], ti[6], ti[2], ti[3], ti[4] - ti[0] - ti[1] - ti[5] - ti[2] - ti[3], &idx); };
    std::vector<uint64_t> dur[NR];
    for (int r = 0; r < NR; ++r) { out[(sl * NR + r) * 4] = ~0ULL; out[(sl * NR + r) * 4 + 1] = 0; out[(sl * NR + r) * 4 + 2] = 0; out[(sl * NR + r) * 4 + 3] = 0; }
    for (unsigned b = 0; b < grid; ++b) {
      const uint64_t t0 = h[((size_t)sl * TIMING_BLOCKS + b) * 2], t1 = h[((size_t)sl * TIMING_BLOCKS + b) * 2 + 1];
      if (!t1) continue;
      const int r = role_of(b);
      out[(sl * NR + r) * 4] = std::min(out[(sl * NR + r) * 4], t0);
      out[(sl * NR + r) * 4 + 1] = std::max(out[(sl * NR + r) * 4 + 1], t1);
      dur[r].push_back(t1 - t0);
    }
    if (std::getenv("TFRA_STEP_OCCUPANCY") && sl == 5 && grid) {   // (tuning) resident blocks per role, every microsecond of one launch
      uint64_t tmin = ~0ULL, tmax = 0;
      for (unsigned b = 0; b < grid; ++b) {
        const uint64_t t0 = h[((size_t)sl * TIMING_BLOCKS + b) * 2], t1 = h[((size_t)sl * TIMING_BLOCKS + b) * 2 + 1];
        if (t1) { tmin = std::min(tmin, t0); tmax = std::max(tmax, t1); }
      }
      for (uint64_t t = tmin; t < tmax; t += 100) {   // (the clock ticks at 100 MHz)
        unsigned n[NR] = {0, 0, 0, 0, 0, 0};
        for (unsigned b = 0; b < grid; ++b) {
          const uint64_t t0 = h[((size_t)sl * TIMING_BLOCKS + b) * 2], t1 = h[((size_t)sl * TIMING_BLOCKS + b) * 2 + 1];
          if (t1 && t0 <= t && t < t1) n[role_of(b)] += 1;
        }
        std::fprintf(stderr, "t %2llu us: build %4u scatter %4u map %4u write-back %4u lookup %4u tail %3u  = %4u blocks\n", (unsigned long long)((t - tmin) / 100), n[0], n[1],
                     n[5], n[2], n[3], n[4], n[0] + n[1] + n[2] + n[3] + n[4] + n[5]);
      }
    }
    for (int r = 0; r < NR; ++r) {
      if (dur[r].empty()) continue;
      std::sort(dur[r].begin(), dur[r].end());
      out[(sl * NR + r) * 4 + 2] = dur[r][dur[r].size() / 2];
      out[(sl * NR + r) * 4 + 3] = dur[r][dur[r].size() * 95 / 100];
    }
    d->tinfo[sl][4] = 0;
  }
  return TFRA_OK;
}

// Measurement: HIP events around the launch of each of the next `steps` overlapped steps (on the stream they are launched on);
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
static bool plan_is_listless(const tfra_sparse_plan* pl) { return pl->kind == 1 && pl->tab_state[pl->set_parity] == 2; }

// variant (TFRA_STEP_VARIANT, tuning): bits 0-2 kernel (0: 8 keys per wave in the write-back, 5 blocks per CU | 1: 4 keys | 2: 4 blocks per CU | 3: 6), 8 every plan as a launch of
// its own, 16 time stamps, 32 no MAP role (round 4's lookup), 64 the lookup reads the table's lines for every id.  TFRA_STEP_OWN_SLICE: a fixed
// write-back slice (else sized from the batch's distinct-key count), TFRA_STEP_FIND_FIRST: lookup blocks in front of the write-back's
static void launch_step(int variant, unsigned grid, hipStream_t s, const StepArgs& a, bool lru) {   // the overlapped step: one launch
  const int k = variant & 7;
  if (!lru) step_k_gen<<<grid, 256, 0, s>>>(a);
  else if (variant & 16) step_k_u2_t<<<grid, 256, 0, s>>>(a);
  else if (k == 1) step_k_u1<<<grid, 256, 0, s>>>(a);
  else if (k == 2) step_k_u2_w4<<<grid, 256, 0, s>>>(a);
  else if (k == 3) step_k_u2_w6<<<grid, 256, 0, s>>>(a);
  else step_k_u2<<<grid, 256, 0, s>>>(a);
}

// One step (n == 0 and no look-ahead: just the pending write-back, the flush).  Caller holds d->t->step_mu.
static int step_overlap_one(tfra_step_driver* d, size_t n, const int64_t* ids, void* rows_out, uint8_t* exists_out, const void* defaults,
                            int default_is_full, const void* values_prev, const uint64_t* scores_prev, size_t n_next,
                            const int64_t* ids_next, size_t n_next2, const int64_t* ids_next2, hipStream_t s) {
  Table* t = d->t;
  if (n && (!ids || !rows_out || !defaults)) return set_error(TFRA_ERR_INVALID, "step_overlap: null buffer");
  if (n > MAX_IDS || n_next > MAX_IDS || n_next2 > MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "step_overlap: at most 2^18 ids per step");
  if (d->pending && !values_prev) return set_error(TFRA_ERR_INVALID, "step_overlap: the previous step's batch has not been written back: values_prev is null");
  if ((n_next && !ids_next) || (n_next2 && !ids_next2)) return set_error(TFRA_ERR_INVALID, "step_overlap: null look-ahead ids");
  constexpr unsigned NPL = tfra_step_driver::NPL;
  const unsigned slot = d->seq % NPL;
  tfra_sparse_plan* plan_cur = d->plans[slot];
  tfra_sparse_plan* plan_prev = d->pending ? d->plans[d->pend_slot] : nullptr;
  tfra_sparse_plan* plan_next = d->plans[(d->seq + 1) % NPL];
  tfra_sparse_plan* plan_next2 = d->plans[(d->seq + 2) % NPL];
  std::unique_lock<std::mutex> lock(t->mu);
  int rc = t->enter(s);
  if (rc) return rc;
  const bool lfu = t->opts.strategy == TFRA_EVICT_LFU;
  // this batch's plan: built by the previous call (look-ahead), else here, in front of the step (one more launch)
  if (n && !(d->ahead && d->ahead_ids == ids && d->ahead_n == n)) {
    rc = setplan_build(plan_cur, n, ids, s, lfu);
    if (rc) return rc;
    d->n_built_in_front += 1;
  }
  if (!n) plan_cur->n = 0;
  d->ahead = false;
  // a MAP list is good for exactly one lookup: these ids, forwarded from that plan as it was built then
  const bool list_ok = d->map_valid && n && d->map_ids == ids && d->map_n == n && plan_prev && d->map_plan == plan_prev && d->map_gen == plan_prev->gen;
  d->map_valid = false;
  // Announced batches are recognised by (address, length).  Whatever this call does not consume is disarmed HERE, on every path: pairs
  // scattered for a batch that is not announced again (no ids_next, the sequential path, a plan launch in front) must not meet a
  // later batch that happens to live at the same address.
  if (!(n_next && plan_next->scat_ids == ids_next && plan_next->scat_n == n_next)) { plan_next->scat_ids = nullptr; plan_next->scat_n = 0; }
  plan_cur->scat_ids = nullptr; plan_cur->scat_n = 0;
  if (plan_prev) { plan_prev->scat_ids = nullptr; plan_prev->scat_n = 0; }
  // (plan_next2's segments are filled by this call or not at all)
  plan_next2->scat_ids = nullptr; plan_next2->scat_n = 0;
  unsigned* tags = t->ensure_own_tags(s);
  // what the TABLE must be for the overlap (constant over its life, but for `dense`) and what this CALL must be
  const bool lru_like = t->opts.strategy == TFRA_EVICT_LRU || t->opts.strategy == TFRA_EVICT_EPOCHLRU;   // a new key is always admitted
  // Round 6: a GROWING table (the cuckoo flavour: no eviction strategy, no max_capacity — what TFRA's default creator instantiates) qualifies
  // too: every key of a batch ends up in it (it grows — own_prepare's prepare_insert, in stream order in front of the launch — instead of
  // refusing or evicting), a new key that finds no free slot in its two home buckets goes to the tail's general (walking) path like any
  // left-over key, and nothing is ever evicted under the lookup.  TFRA_STEP_GROWING=0 keeps such tables on the sequential path.
  static const bool growing_ok = [] { const char* e = std::getenv("TFRA_STEP_GROWING"); return !e || std::atoi(e) != 0; }();
  const bool growing = growing_ok && t->opts.strategy == TFRA_EVICT_NONE && t->opts.max_capacity == 0;
  // ... and so does a BOUNDED LRU / EPOCHLRU table that is not (yet) at its max_capacity or not yet dense: below max_capacity it grows like
  // the cuckoo flavour; at max_capacity but sparse a new key walks four buckets and an eviction — if it comes to one — is the tail's (after
  // every write-back block, victims the lookup asked for corrected as always).  An Hkv table whose max_capacity is never reached — a common
  // deployment — used to stay on the sequential path for ever (reasons 16 / 32).
  const bool any_fill = growing_ok && lru_like;
  const unsigned why_table = (tags ? 0u : 4u) | ((t->opts.aux_fields == 0 && (lru_like || growing)) ? 0u : 8u) |
                             ((growing || any_fill || t->at_max_capacity()) ? 0u : 16u) | ((growing || any_fill || t->dense) ? 0u : 32u) |
                             (t->capture_safe ? 128u : 0u) | ((t->field_bytes & 15u) ? 2u : 0u);
  const bool aligned = (((uintptr_t)rows_out | (uintptr_t)defaults | (uintptr_t)values_prev) & 15) == 0;
  const unsigned why = why_table | (aligned ? 0u : 2u) | (scores_prev ? 8u : 0u) | ((!plan_prev || plan_prev->n > 0) ? 0u : 64u);
  const bool eligible = why == 0 && (n > 0 || plan_prev);
  const unsigned step = ++d->step_no;
  if (!eligible) {
    if (n || plan_prev) d->why_sequential = why ? why : 1u;
    // the same results one after the other: write-back of the previous batch, this lookup, the next batch's plan
    if (plan_prev && plan_prev->n) {
      if (plan_is_listless(plan_prev)) {   // its plan has no key list (built by a step launch): build it again, with one
        rc = setplan_build(plan_prev, d->pend_n, d->pend_ids, s, lfu);
        if (rc) return rc;
      }
      rc = upsert_planned_impl(d->tp, plan_prev, values_prev, scores_prev, s, nullptr, 0);
      if (rc) return rc;
    }
    lock.unlock();
    if (n) { rc = tfra_table_find(d->tp, n, ids, rows_out, exists_out, defaults, default_is_full, s); if (rc) return rc; }
    plan_next->scat_ids = nullptr; plan_next->scat_n = 0;
    if (n_next) {
      rc = setplan_build(plan_next, n_next, ids_next, s, lfu);
      if (rc) return rc;
      d->n_built_in_front += 1;
    }
    if (n || plan_prev) d->n_sequential += 1;
  } else {
    StepArgs a{};
    OwnLaunch L{};
    if (plan_prev) {
      rc = own_prepare(t, plan_prev, values_prev, nullptr, s, nullptr, &L);
      if (rc) return rc;
      a.own = L.a;
      a.ctr = L.ctr; a.own_gen = L.og;
      a.fwd = probe_of(plan_prev);
      if (plan_is_listless(plan_prev) && plan_prev->ucnt) { a.own_ucnt = plan_prev->ucnt; a.own_ucnt_n = plan_prev->set_m2 / SET_WIN; }
      // ~40 keys per write-back block (two rounds of its four waves): the slice follows the density of the plan's table, known from
      // the distinct-key count an earlier launch's tail left in pinned memory (0: none yet)
      a.own_slice = d->own_slice;
      if (!d->own_slice_fixed) {
        const unsigned u_est = __atomic_load_n(d->progress + 1, __ATOMIC_RELAXED);
        // (a launch whose lookup is SMALL — the distinct ids one rank of a sharded table serves: 22 K lookups beside 22 K writes, the whole
        // grid resident at once — runs faster with ~64 keys per write-back block: slices of 640-768 slots 26 us, 448 slots 29 us;
        // the metric's full batches keep ~40: scripts/sweep_owner.sh)
        const bool small_lookup = n < 65536;
        if (u_est) a.own_slice = std::min(small_lookup ? 768u : 512u, std::max(96u, (unsigned)((small_lookup ? 64ull : 40ull) * (a.fwd.m2 + 2) / u_est) & ~31u));
      }
      a.own_blocks = (a.fwd.m2 + 2 + a.own_slice - 1) / a.own_slice;
    } else {
      a.own.v = t->view_of(t->cur);
      a.own_blocks = 0;
      a.fwd = SetProbe{d->dummy, 4};
    }
    a.progress = d->progress; a.progress_val = step; a.stat = d->stat; a.tbuf = d->tbuf;
    a.patch_keys = reinterpret_cast<i64*>(d->patch + 256) + (size_t)PATCH_GCAP * (step & 1u);
    a.nxt = n ? probe_of(plan_cur) : SetProbe{d->dummy, 4};
    a.n = (unsigned)n; a.ids = (const i64*)ids; a.out = (unsigned char*)rows_out; a.exists = exists_out;
    a.defaults = (const unsigned char*)defaults; a.full = default_is_full;
    a.find_blocks = (unsigned)((n + 63) / 64);
    a.tail_blocks = plan_prev ? TAIL_BLOCKS : 0u;
    a.sync = reinterpret_cast<unsigned*>(d->patch + 256 + 2 * PATCH_GCAP * 8 + 1280 * (step & 1u));
    a.sync_next = reinterpret_cast<unsigned*>(d->patch + 256 + 2 * PATCH_GCAP * 8 + 1280 * ((step & 1u) ^ 1u));
    a.patch_count = a.sync + 32 * 9 + 1; a.patch_count_next = a.sync_next + 32 * 9 + 1;   // (read with the tail's arrivals as one 8-byte word)
    a.zero4 = plan_prev ? reinterpret_cast<unsigned*>(L.next_ctr) : nullptr;
    a.serial_probe = (d->variant & 64) ? 0 : 1;
    a.ablate = (d->ablate && d->step_no >= d->ablate_after) ? d->ablate : 0;   // (tuning: TFRA_STEP_ABLATE / _AFTER; results are wrong)
    int map_slot_new = -1;
    if (list_ok) {
      a.find_list = reinterpret_cast<const uint4*>(d->mapbuf + (size_t)d->map_slot * d->map_cap * 16);
      d->n_find_listed += 1;
      a.find_blocks = (unsigned)((n + MAP_SEG - 1) / MAP_SEG) * (MAP_SEG / 64u);   // chunk-major over whole segments
      a.find_first = d->find_first > 0 ? std::min((unsigned)d->find_first, a.find_blocks) : 0u;
    }
    // the NEXT lookup's positions, probed in this batch's plan (complete before this launch): the MAP role
    if (n && n_next && !(d->variant & 32)) {
      if (d->map_cap < std::max(n_next, (size_t)4096)) {
        if (d->mapbuf) { if (hipDeviceSynchronize() != hipSuccess || hipFree(d->mapbuf) != hipSuccess) return set_error(TFRA_ERR_HIP, "step_overlap: free"); d->mapbuf = nullptr; }
        if (a.find_list) { a.find_list = nullptr; a.find_blocks = (unsigned)((n + 63) / 64); a.find_first = 0; }   // (it lived in the buffer just freed; n <= the old capacity < n_next)
        size_t cap = 4096;
        while (cap < n_next) cap <<= 1;
        if (hipMalloc((void**)&d->mapbuf, 2 * cap * 16) != hipSuccess) { d->mapbuf = nullptr; d->map_cap = 0; return set_error(TFRA_ERR_OOM, "step_overlap: hipMalloc"); }
        d->map_cap = cap;
      }
      const unsigned ms = a.find_list ? (d->map_slot ^ 1u) : 0u;
      a.map_n = (unsigned)n_next; a.map_ids = (const i64*)ids_next; a.map_blocks = (unsigned)((n_next + MAP_SEG - 1) / MAP_SEG);
      a.map_out = reinterpret_cast<uint4*>(d->mapbuf + (size_t)ms * d->map_cap * 16);
      map_slot_new = (int)ms;   // (the driver's record of the list is made behind the launch: an error return in between must not arm it)
    }
    // the next batch's plan: its pairs were scattered by the previous call's launch -> this launch builds the table; else a launch of its own, in front
    if (n_next) {
      if (plan_next->scat_ids == ids_next && plan_next->scat_n == n_next && !(d->variant & 8)) {
        const SetTab tb = setplan_take_listless(plan_next, n_next);
        a.build_ent = tb.ent; a.build_m2 = plan_next->set_m2; a.build_tiles = plan_next->seg_tiles; a.build_blocks = plan_next->set_m2 / SET_WIN;
        a.build_pairs = plan_next->seg_pairs; a.build_cnt = plan_next->seg_cnt; a.build_ovf = plan_next->ovf_pairs;
        a.build_ovf_cnt = plan_next->ovf_cnt + 32 * (plan_next->scat_use & 1u);
        a.build_ucnt = plan_next->ucnt;
        d->n_built_in_launch += 1;
      } else {
        rc = setplan_build(plan_next, n_next, ids_next, s, false);
        if (rc) return rc;
        d->n_built_in_front += 1;
      }
      plan_next->scat_ids = nullptr; plan_next->scat_n = 0;
    }
    // the batch after next: its distinct (id, last position) pairs go into plan_next2's segments
    if (n_next2 && !(d->variant & 8)) {
      rc = setplan_prepare_listless(plan_next2, n_next2, s);
      if (rc) return rc;
      plan_next2->scat_use += 1;
      a.scat_n = (unsigned)n_next2; a.scat_ids = (const i64*)ids_next2; a.scat_m2 = plan_next2->set_m2;
      a.scat_tiles = (unsigned)((n_next2 + 1023) / 1024); a.scat_blocks = a.scat_tiles;
      a.scat_pairs = plan_next2->seg_pairs; a.scat_cnt = plan_next2->seg_cnt; a.scat_ovf = plan_next2->ovf_pairs;
      a.scat_ovf_cnt = plan_next2->ovf_cnt + 32 * (plan_next2->scat_use & 1u);
      a.scat_ovf_cnt_next = plan_next2->ovf_cnt + 32 * ((plan_next2->scat_use & 1u) ^ 1u);
      plan_next2->seg_tiles = a.scat_tiles; plan_next2->scat_ids = ids_next2; plan_next2->scat_n = n_next2;
    }
    if (plan_prev && d->last_tail_step + 1 != step) {   // the launch before had no tail: nobody has zeroed this launch's counters
      if (hipMemsetAsync(a.patch_count, 0, 4, s) != hipSuccess || hipMemsetAsync(a.sync, 0, 1280, s) != hipSuccess) return set_error(TFRA_ERR_HIP, "step_overlap: memset");
    }
    if (plan_prev) d->last_tail_step = step;
    const bool timed = d->kev_left > 0 && plan_prev;
    if (timed) (void)hipEventRecord(d->kev[d->kev_used * 3], s);
    {
      // ONE launch: the lookup runs beside the write-back (forwarding, deferred evictions, corrections).  (Measured against it, on
      // the metric's configuration: the same roles as TWO launches one after the other — write-back + tail, then lookup + plan
      // builders, no forwarding — 52 us per step against 33: the write-back alone in its launch still takes 25 us.)
      const unsigned grid = a.build_blocks + a.scat_blocks + a.map_blocks + a.own_blocks + a.find_blocks + a.tail_blocks;
      if (d->tbuf) { unsigned* ti = d->tinfo[step % TIMING_SLOTS]; ti[0] = a.build_blocks; ti[1] = a.scat_blocks; ti[2] = a.own_blocks; ti[3] = a.find_blocks; ti[4] = grid; ti[5] = a.map_blocks; ti[6] = a.find_first; }
      launch_step(d->variant, grid, s, a, t->opts.strategy == TFRA_EVICT_LRU);
      if (timed) (void)hipEventRecord(d->kev[d->kev_used * 3 + 1], s);
    }
    if (timed) { (void)hipEventRecord(d->kev[d->kev_used * 3 + 2], s); d->kev_used += 1; d->kev_left -= 1; }
    if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "step_overlap: launch failed");
    if (map_slot_new >= 0) {
      d->map_slot = (unsigned)map_slot_new; d->map_valid = true; d->map_ids = ids_next; d->map_n = n_next; d->map_plan = plan_cur; d->map_gen = plan_cur->gen;
    }
    if (plan_prev) step_epoch_public(t);
    d->n_overlapped += 1;
  }
  d->pending = n > 0;
  d->pend_slot = slot; d->pend_ids = ids; d->pend_n = n;
  if (n_next) { d->ahead = true; d->ahead_ids = ids_next; d->ahead_n = n_next; }
  if (n) d->seq += 1;
  return TFRA_OK;
}

extern "C" int tfra_table_step_overlap(tfra_step_driver_t* d, size_t n, const int64_t* ids, void* rows_out, uint8_t* exists_out,
                                       const void* defaults, int default_is_full, const void* values_prev, const uint64_t* scores_prev,
                                       size_t n_next, const int64_t* ids_next, size_t n_next2, const int64_t* ids_next2, tfra_stream_t stream) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_overlap: null driver");
  if (!n) return set_error(TFRA_ERR_INVALID, "step_overlap: empty batch (tfra_table_step_overlap_flush writes a pending batch back)");
  std::lock_guard<std::mutex> step_lock(d->t->step_mu);
  return step_overlap_one(d, n, ids, rows_out, exists_out, defaults, default_is_full, values_prev, scores_prev, n_next, ids_next, n_next2, ids_next2,
                          (hipStream_t)stream);
}

extern "C" int tfra_table_steps_overlap(tfra_step_driver_t* d, size_t count, const tfra_overlap_step* steps, tfra_stream_t stream) {
  if (!d || (count && !steps)) return set_error(TFRA_ERR_INVALID, "steps_overlap: null argument");
  std::lock_guard<std::mutex> step_lock(d->t->step_mu);
  for (size_t i = 0; i < count; ++i) {
    const tfra_overlap_step& q = steps[i];
    if (q.struct_size != sizeof(tfra_overlap_step)) return set_error(TFRA_ERR_INVALID, "steps_overlap: struct_size mismatch");
    if (!q.n) return set_error(TFRA_ERR_INVALID, "steps_overlap: empty batch");
    int rc = step_overlap_one(d, q.n, q.ids, q.rows_out, q.exists_out, q.defaults, q.default_is_full, q.values_prev, q.scores_prev, q.n_next,
                              q.ids_next, q.n_next2, q.ids_next2, (hipStream_t)stream);
    if (rc) return rc;
  }
  return TFRA_OK;
}

extern "C" int tfra_table_step_overlap_flush(tfra_step_driver_t* d, const void* values_prev, const uint64_t* scores_prev, tfra_stream_t stream) {
  if (!d) return set_error(TFRA_ERR_INVALID, "step_overlap_flush: null driver");
  std::lock_guard<std::mutex> step_lock(d->t->step_mu);
  if (!d->pending) return TFRA_OK;
  if (!values_prev) return set_error(TFRA_ERR_INVALID, "step_overlap_flush: null values_prev");
  // the pending write-back alone: a step without a lookup (its launch holds the ownership pass only), or the planned upsert
  return step_overlap_one(d, 0, nullptr, nullptr, nullptr, nullptr, 0, values_prev, scores_prev, 0, nullptr, 0, nullptr, (hipStream_t)stream);
}

#endif  // TFRA_STEP_HOST_PART
