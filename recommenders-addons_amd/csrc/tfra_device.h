// Device-side layout and probe primitives of the MI355X dynamic-embedding table.
//
// Layout in HBM: ONE allocation of nb bucket blocks + 2 side rows.
//   bucket block b (bucket_stride bytes, at base + b*bucket_stride):
//     +0    key line   : 16 x int64 = ONE 128-byte line = 15 key slots + 1 meta word (two
//                        monotone overflow flags, see META_OVF0/1).  A probe is one coalesced 128-B read by the 16
//                        lanes of a key group; 4 key groups per wave64.
//     +128  score line : 16 x uint64 (only with an eviction strategy; hdr = 256, else hdr = 128)
//     +hdr  15 rows x row_stride bytes, row_stride = 16-B multiple of (1+aux_fields)*dim*sizeof(V):
//                        [embedding | slot1 | slot2 ...] co-located so a fused optimizer touches
//                        one contiguous segment.
//   A key's line, its score and its row live in the SAME block (4096 B for 256-B rows with scores).
//   (Measured, scripts/mb/tlb_probe.hip: on a 264-GiB allocation a dependent row read costs the same in the
//   same 4-KiB block as anywhere else — address translation does not bound these kernels; instruction issue
//   and the number of dependent round trips do.  The block layout is kept because it costs nothing.)
//   side rows: 2 rows behind the last block = store for the two key values used as sentinels.
//
// Probe sequence of key k: b0 = mulhi(fmix64(k), nb); b1 = mulhi(fmix64(h^C), nb) (!= b0);
// then b1+1, b1+2, ... (mod nb).  First-fit insertion + the monotone overflow flags mean a find
// stops at the first bucket that holds the key or is not flagged: 1 line for nearly every
// key at load factor <= 0.5, no tombstones (erase simply empties the slot).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfra {

typedef long long i64;
typedef unsigned long long u64;

constexpr i64 EMPTY_KEY = (i64)0x8000000000000000ULL;   // INT64_MIN
constexpr i64 LOCKED_KEY = (i64)0x8000000000000001ULL;  // slot being replaced (eviction)
constexpr int SLOTS = 15;                               // key slots per 128-B bucket line
constexpr int NUM_RESERVED = 2;                         // side rows for EMPTY_KEY / LOCKED_KEY
// meta word (word 15 of the key line): two monotone flags.  OVF0 = a key whose FIRST bucket (b0) is this one was
// placed elsewhere (its b0 was full); OVF1 = a key passed through this bucket further down its sequence (as b1,
// b1+1, ...) because it was full.  A search for key k stops at b0 unless OVF0(b0), and at any later bucket of its
// sequence unless OVF1 of that bucket: a bucket being full of OTHER keys' first choices does not lengthen k's search.
// (With a single flag a bounded table running at capacity has every bucket flagged — each is some key's b0 —
// and a miss walks on and on.)
constexpr u64 META_OVF0 = 1ULL, META_OVF1 = 2ULL;
constexpr int SIZE_SHARDS = 256;       // size counter sharded over 256 lines (one per channel-ish)
constexpr int SIZE_SHARD_STRIDE = 16;  // u64 words between shards (128 B)

struct TableView {
  unsigned char* base;   // bucket block b at base + b*bucket_stride
  u64 nb;
  u64 bucket_stride;     // hdr + 15*row_stride
  unsigned hdr;          // 128 (key line) or 256 (key line + score line)
  unsigned field_bytes;  // dim*sizeof(V)
  unsigned row_stride;   // bytes between rows
  unsigned n_fields;     // 1 + aux_fields
  unsigned* reserved_present;  // [NUM_RESERVED]
  u64* size_shards;            // [SIZE_SHARDS*SIZE_SHARD_STRIDE], wrapping signed deltas
  int* winner;                 // [nb*15+2] scratch for duplicate resolution (may be null)
  unsigned* err_count;         // keys that could not be placed (table full)
  const unsigned* dense_flag;  // device word, != 0 once a bounded table at max_capacity holds > 60 % of its slots
                               // (set by density_kernel in stream order: the host's view of the size lags by
                               // however many calls are queued, this one does not)
};

__device__ __forceinline__ bool has_scores(const TableView& v) { return v.hdr > 128; }
// Integer instruction count, not HBM bandwidth, bounds these kernels (a wave64 VALU instruction takes 4 cycles, a
// 32-bit multiply 16: find_kernel was ~1000 instructions per 16 keys, a quarter of its cycles in 64-bit multiplies),
// so the address arithmetic is kept to ONE v_mad_u64_u32: bucket indices and the bucket stride are < 2^32
// (tfra_table_create rejects more than 2^32 - 1 buckets).
__device__ __forceinline__ unsigned char* bucket_ptr(const TableView& v, u64 b) {
  return v.base + (u64)(unsigned)b * (u64)(unsigned)v.bucket_stride;
}
__device__ __forceinline__ i64* key_line(const TableView& v, u64 b) { return reinterpret_cast<i64*>(bucket_ptr(v, b)); }
__device__ __forceinline__ u64* score_line(const TableView& v, u64 b) { return reinterpret_cast<u64*>(bucket_ptr(v, b) + 128); }
// "word" index w = b*16 + slot names one key slot (and its score)
__device__ __forceinline__ i64* key_word(const TableView& v, u64 w) { return key_line(v, w >> 4) + (w & 15); }
__device__ __forceinline__ u64* score_word(const TableView& v, u64 w) { return score_line(v, w >> 4) + (w & 15); }
__device__ __forceinline__ unsigned char* row_at(const TableView& v, u64 b, unsigned slot) {
  return bucket_ptr(v, b) + (v.hdr + slot * v.row_stride);
}
// row index r = b*15 + slot (< 2^36) -> (b, slot) without a 64-bit division: r = hi*2^32 + lo with hi < 16 and
// 2^32 = 15*286331153 + 1, so r = 15*(hi*286331153) + (hi + lo); the rest (< 2^32 + 16) divides by the usual
// 32-bit magic (0x88888889 >> 35, exact below 2^35 / 7).
__device__ __forceinline__ void split_row(u64 r, u64& b, unsigned& slot) {
  const unsigned hi = (unsigned)(r >> 32);
  const u64 rest = (u64)hi + (u64)(unsigned)r;
  const unsigned q = (unsigned)((rest * 0x88888889ULL) >> 35);
  b = (u64)hi * 286331153ULL + q;
  slot = (unsigned)(rest - (u64)q * 15);
}
// row index r = b*15 + slot; r >= nb*15 = side rows
__device__ __forceinline__ unsigned char* row_ptr(const TableView& v, i64 row) {
  u64 b;
  unsigned slot;
  split_row((u64)row, b, slot);
  if (b >= v.nb) return bucket_ptr(v, v.nb) + (size_t)((u64)row - v.nb * 15) * v.row_stride;
  return row_at(v, b, slot);
}

__device__ __forceinline__ u64 fmix64(u64 k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}
// Two home buckets from ONE 64-bit mix: b0 = the high word of fmix64(key) range-reduced by a 32-bit multiply-high
// (nb < 2^32), b1 = the low word re-mixed (murmur3 fmix32) and reduced the same way — 10 32-bit multiplies per key
// instead of the 26 of two fmix64 + two 64-bit multiply-highs.
__device__ __forceinline__ unsigned fmix32(unsigned x) {
  x ^= x >> 16; x *= 0x85ebca6bu;
  x ^= x >> 13; x *= 0xc2b2ae35u;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ u64 bucket0(i64 key, u64 nb, u64& h) {
  h = fmix64((u64)key);
  return (u64)__umulhi((unsigned)(h >> 32), (unsigned)nb);
}
__device__ __forceinline__ u64 bucket1(u64 h, u64 b0, u64 nb) {
  unsigned b1 = __umulhi(fmix32((unsigned)h ^ 0x9e3779b9u), (unsigned)nb);
  if (b1 == (unsigned)b0) b1 = (b1 + 1 == (unsigned)nb) ? 0u : b1 + 1;
  return (u64)b1;
}
__device__ __forceinline__ u64 next_bucket(u64 b, u64 nb) { return (b + 1 == nb) ? 0 : b + 1; }
__device__ __forceinline__ bool is_reserved_key(i64 k) { return k <= LOCKED_KEY; }
__device__ __forceinline__ int reserved_index(i64 k) { return (int)(k - EMPTY_KEY); }

// Keys are read with agent-scope relaxed atomics (sc1: served by L2, never a stale L1 line) in
// every kernel that can race with slot claims of the same launch.
__device__ __forceinline__ i64 load_key_coherent(const i64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ i64 shfl_i64(i64 v, int src) {
  int lo = __shfl((int)(u64)v, src), hi = __shfl((int)((u64)v >> 32), src);
  return (i64)(((u64)(unsigned)hi << 32) | (unsigned)lo);
}

// "All of these are live in registers here": an empty asm that takes every value as an in/out
// operand.  Placed after a group of independent loads it makes the compiler issue ALL of them
// before the first use (one s_waitcnt for the group) instead of pairing each load with its
// consumer — hipcc otherwise serialises e.g. load/store pairs of a row copy (4 latencies instead
// of 1; measured 23-35 us vs 11 us for find_kernel).  __builtin_amdgcn_sched_barrier was tried
// for the same purpose and made the kernel 3x slower.
__device__ __forceinline__ void keep_live(i64& a, i64& b, i64& c, i64& d) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void keep_live(uint4& a, uint4& b, uint4& c, uint4& d) {
  asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w),
                    "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
}
__device__ __forceinline__ void keep_live(float4& a, float4& b, float4& c, float4& d) {
  asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w),
                    "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
}
__device__ __forceinline__ void keep_live(uint2& a, uint2& b, uint2& c, uint2& d) {
  asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y), "+v"(c.x), "+v"(c.y), "+v"(d.x), "+v"(d.y));
}
__device__ __forceinline__ void keep_live(unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// 1- and 2-byte granules (odd row sizes) are not worth pinning
__device__ __forceinline__ void keep_live(unsigned short&, unsigned short&, unsigned short&, unsigned short&) {}
__device__ __forceinline__ void keep_live(unsigned char&, unsigned char&, unsigned char&, unsigned char&) {}

// Write-through 16-B store (sc0 sc1): the line leaves the XCD's L2 while the kernel is still running
// instead of staying dirty until the kernel boundary (MI355X_MICROARCH.md: a dependent kernel
// boundary costs + B / 6 TB/s for B dirty bytes — 5.6 us behind find's 33.5 MB of output).
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt16(void* p, uint4 v) {
  v4u_t x = {v.x, v.y, v.z, v.w};
  // s_nop 1: a store of more than 64 bits reads its data registers after issue — a VALU write to them needs two wait states
  // behind it, and the compiler's hazard recogniser does not look into inline assembly (a loop that recomputed the stored
  // float4 right after the store wrote the NEXT iteration's values: the partial sums of rows wider than 64 floats)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}

// ---- find: continue a probe whose first line `k` (bucket b) is already in registers --------
// Returns the row index (b*15+slot) or -1.  All 16 lanes of the group call with the same key.
template <bool COHERENT>
__device__ __forceinline__ i64 probe_find_from(const TableView& v, i64 key, u64 h, u64 b, i64 k,
                                               int sub, int gshift, const i64* k_second = nullptr) {
  if (is_reserved_key(key)) {
    int r = reserved_index(key);
    return v.reserved_present[r] ? (i64)(v.nb * SLOTS + r) : -1;
  }
  u64 b1 = bucket1(h, b, v.nb);
  for (u64 step = 0;; ++step) {
    u64 m = __ballot(sub < SLOTS && k == key);
    unsigned hit = (unsigned)(m >> gshift) & 0x7fffu;
    if (hit) return (i64)(b * SLOTS + (__ffs(hit) - 1));
    i64 meta = shfl_i64(k, gshift + 15);
    if (!((u64)meta & (step == 0 ? META_OVF0 : META_OVF1)) || step >= v.nb) return -1;
    b = (step == 0) ? b1 : next_bucket(b, v.nb);
    if (step == 0 && k_second) k = *k_second;  // b1's line, preloaded by the caller
    else k = COHERENT ? load_key_coherent(key_line(v, b) + sub) : key_line(v, b)[sub];
  }
}

// Same search with both home buckets already hashed by the caller (one lane per key hashes, the group gets b0 / b1
// by shuffle: a quarter of the hash instructions), returning the slot as a WORD index b*16 + slot (side rows:
// nb*16 + r) so that the row address needs no division; -1 = absent.
__device__ __forceinline__ i64 probe_find_word(const TableView& v, i64 key, u64 b0, u64 b1, i64 k, int sub, int gshift,
                                               const i64* k_second = nullptr) {
  if (is_reserved_key(key)) {
    int r = reserved_index(key);
    return v.reserved_present[r] ? (i64)(v.nb * 16 + r) : -1;
  }
  u64 b = b0;
  for (u64 step = 0;; ++step) {
    u64 m = __ballot(sub < SLOTS && k == key);
    unsigned hit = (unsigned)(m >> gshift) & 0x7fffu;
    if (hit) return (i64)(b * 16 + (__ffs(hit) - 1));
    i64 meta = shfl_i64(k, gshift + 15);
    if (!((u64)meta & (step == 0 ? META_OVF0 : META_OVF1)) || step >= v.nb) return -1;
    b = (step == 0) ? b1 : next_bucket(b, v.nb);
    if (step == 0 && k_second) k = *k_second;  // b1's line, preloaded by the caller
    else k = key_line(v, b)[sub];
  }
}
// row of a word index (see probe_find_word); side rows sit behind the last bucket block, without a header
__device__ __forceinline__ unsigned char* word_row_ptr(const TableView& v, u64 word) {
  const u64 b = word >> 4;
  const unsigned slot = (unsigned)word & 15u;
  return bucket_ptr(v, b) + (slot * v.row_stride + (b >= v.nb ? 0u : v.hdr));
}

template <bool COHERENT>
__device__ __forceinline__ i64 probe_find(const TableView& v, i64 key, int sub, int gshift) {
  u64 h;
  u64 b = bucket0(key, v.nb, h);
  i64 k = COHERENT ? load_key_coherent(key_line(v, b) + sub) : key_line(v, b)[sub];
  return probe_find_from<COHERENT>(v, key, h, b, k, sub, gshift);
}

// ---- insert: find the key or claim the FIRST empty slot of its probe sequence -------------
// Within one launch slots only go EMPTY -> key, and every inserter takes the first empty slot
// in probe order with a CAS, so two groups inserting the same key can never end up in two
// different slots.  Returns row index, or -1 when no slot could be found.
// `k_first` = the key's first bucket line (b0), already loaded by the caller (coherently) so that
// a kernel can put the first probes of several keys in flight before resolving any of them.
// bounded != 0: the table cannot grow (Hkv flavour at max_capacity).  A NEW key is placed within the first 4
// buckets of its sequence (b0, b1, b1+1, b1+2 — at load factor 0.5 the chance that all four are full is ~1e-8, so
// nothing is evicted before the table is really filling up: T/hkv_hashtable_ops_test.py:572-625 pins that);
// bounded == 2 ("dense": the host saw > 60 % of the slots in use): only within its two home buckets b0 / b1
// (HKV likewise confines a key to its one bucket and evicts inside it), so that the OVF1 flags stop spreading and
// a miss on a table running at capacity costs two lines, not an ever longer walk.  When neither the key nor an
// empty slot is found the function returns NEED_EVICT (-2) and the caller replaces the minimum-score entry of the
// two home buckets.  `k_second` (optional) = b1's line, preloaded by callers that expect to need it: both home
// buckets are then in flight together instead of one round trip after the other.
constexpr i64 NEED_EVICT = -2;
__device__ __forceinline__ i64 locate_or_claim_from(const TableView& v, i64 key, u64 h, u64 b0, i64 k_first,
                                                    int sub, int gshift, bool& is_new, int bounded = 0,
                                                    const i64* k_second = nullptr) {
  is_new = false;
  if (is_reserved_key(key)) {
    int r = reserved_index(key);
    unsigned old = 0;
    if (sub == 0) old = atomicExch(&v.reserved_present[r], 1u);
    old = __shfl(old, gshift);
    is_new = (old == 0);
    return (i64)(v.nb * SLOTS + r);
  }
  const u64 b1 = bucket1(h, b0, v.nb);
  if (bounded == 1 && *v.dense_flag) bounded = 2;  // uniform scalar load; see TableView::dense_flag
  for (int attempt = 0; attempt < 1024; ++attempt) {
    u64 b = b0;
    i64 fe = -1;  // word index (b*16+slot) of the first empty slot seen
    for (u64 step = 0; step <= v.nb; ++step) {
      i64 k;
      if (attempt == 0 && step == 0) k = k_first;
      else if (attempt == 0 && step == 1 && k_second) k = *k_second;
      else k = load_key_coherent(key_line(v, b) + sub);
      u64 m = __ballot(sub < SLOTS && k == key);
      unsigned hit = (unsigned)(m >> gshift) & 0x7fffu;
      if (hit) return (i64)(b * SLOTS + (__ffs(hit) - 1));
      u64 e = __ballot(sub < SLOTS && k == EMPTY_KEY);
      unsigned emp = (unsigned)(e >> gshift) & 0x7fffu;
      if (fe < 0 && emp) fe = (i64)(b * 16 + (__ffs(emp) - 1));
      i64 meta = shfl_i64(k, gshift + 15);
      const u64 flag = step == 0 ? META_OVF0 : META_OVF1;
      if (!((u64)meta & flag)) {
        if (fe >= 0) break;  // the key cannot live further along; claim the first empty slot
        if (bounded && step >= (bounded > 1 ? 1u : 3u)) return NEED_EVICT;
        // bucket full and never overflowed: extend the chain through it
        if (sub == 15) atomicOr((u64*)(key_line(v, b) + 15), flag);
      }
      b = (step == 0) ? b1 : next_bucket(b, v.nb);
    }
    if (fe < 0) return bounded ? NEED_EVICT : -1;
    i64 old = 0;
    if (sub == 0) old = (i64)atomicCAS((u64*)key_word(v, (u64)fe), (u64)EMPTY_KEY, (u64)key);
    old = shfl_i64(old, gshift);
    u64 bb = (u64)fe >> 4;
    i64 row = (i64)(bb * SLOTS + ((u64)fe & 15));
    if (old == EMPTY_KEY) { is_new = true; return row; }
    if (old == key) return row;  // an identical key won the race for this very slot
    // another key took it: rescan
  }
  return -1;
}

__device__ __forceinline__ i64 locate_or_claim(const TableView& v, i64 key, int sub, int gshift,
                                               bool& is_new) {
  u64 h;
  const u64 b0 = bucket0(key, v.nb, h);
  i64 k = load_key_coherent(key_line(v, b0) + sub);
  return locate_or_claim_from(v, key, h, b0, k, sub, gshift, is_new);
}

// Victim choice among the 30 slots of (b0, b1) from their key lines kk2[] and score lines sc2[] (lane `sub` holds word
// `sub` of each): minimum score, lowest slot on ties, b0 before b1; an EMPTY slot counts as score 0; LOCKED slots are
// not candidates.  ALU only (shuffles).  best_score == ~0 when nothing is eligible.
__device__ __forceinline__ void select_victim(u64 b0, u64 b1, const i64 (&kk2)[2], const i64 (&sc2)[2], int sub, int gshift,
                                              u64& best_score, u64& best_word, i64& best_key) {
  best_score = ~0ULL; best_word = 0; best_key = 0;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const u64 b = which ? b1 : b0;
    const i64 k = kk2[which];
    u64 sc = (u64)sc2[which];
    const bool cand = sub < SLOTS && k != LOCKED_KEY;
    if (k == EMPTY_KEY) sc = 0;  // an empty slot (erase since the first phase) beats any victim
    u64 my = cand ? sc : ~0ULL;
    u64 packed_lo = (u64)sub;  // 16-lane min with the lane index as tie-break
    for (int o = 8; o > 0; o >>= 1) {
      u64 os = ((u64)(unsigned)__shfl_xor((int)(my >> 32), o) << 32) | (unsigned)__shfl_xor((int)my, o);
      u64 ol = (u64)(unsigned)__shfl_xor((int)packed_lo, o);
      if (os < my || (os == my && ol < packed_lo)) { my = os; packed_lo = ol; }
    }
    const i64 kk = shfl_i64(k, gshift + (int)packed_lo);
    if (my < best_score) { best_score = my; best_word = b * 16 + packed_lo; best_key = kk; }
  }
}

// The same choice with ONE 16-lane reduction: every lane first takes the better of its b0 and b1 candidates (index
// which*16 + sub: b0's slots order before b1's, the lower index wins ties — the order select_victim produces).
__device__ __forceinline__ void select_victim_merged(u64 b0, u64 b1, const i64 (&kk2)[2], const i64 (&sc2)[2], int sub,
                                                     int gshift, u64& best_score, u64& best_word, i64& best_key) {
  u64 s0 = (u64)sc2[0], s1 = (u64)sc2[1];
  if (kk2[0] == EMPTY_KEY) s0 = 0;
  if (kk2[1] == EMPTY_KEY) s1 = 0;
  if (sub >= SLOTS || kk2[0] == LOCKED_KEY) s0 = ~0ULL;
  if (sub >= SLOTS || kk2[1] == LOCKED_KEY) s1 = ~0ULL;
  u64 my = s0;
  unsigned idx = (unsigned)sub;
  if (s1 < s0) { my = s1; idx = 16u + (unsigned)sub; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const u64 os = ((u64)(unsigned)__shfl_xor((int)(my >> 32), o) << 32) | (unsigned)__shfl_xor((int)my, o);
    const unsigned oi = (unsigned)__shfl_xor((int)idx, o);
    if (os < my || (os == my && oi < idx)) { my = os; idx = oi; }
  }
  best_score = my;
  const int src = gshift + (int)(idx & 15u);
  const i64 ka = shfl_i64(kk2[0], src), kb = shfl_i64(kk2[1], src);
  best_key = idx >= 16u ? kb : ka;
  best_word = (idx >= 16u ? b1 : b0) * 16 + (idx & 15u);
}

// ---- eviction (Hkv strategies, table full): replace the minimum-score entry among the 30 slots
// of the key's two home buckets — HKV's "in-bucket min-score eviction" (SURVEY.md appendix D;
// behaviour pinned by T/hkv_hashtable_evict_test.py).  admit_always: LRU-type scores (a new key is
// the most recent); otherwise the new key enters only if in_score >= the minimum score.
// Returns the row whose key word now holds LOCKED_KEY (the caller writes row + score, then publishes
// the key with publish_key), -1 when the key was not admitted, -3 when no victim could be taken.
// victim_key (optional): receives the replaced key.  given_back / n_given_back (optional): the keys whose slot this call locked
// and gave back (up to 4 are kept; the count goes on).
// pre_k / pre_s (optional): key and score lines of (b0, b1) preloaded by the caller — a kernel that handles several keys
// per group puts all of their lines in flight before resolving any (first attempt only; retries reload).
__device__ __forceinline__ i64 evict_and_lock(const TableView& v, i64 key, u64 in_score, bool admit_always, int sub,
                                              int gshift, u64* victim_word, bool& claimed_empty,
                                              const i64* pre_k = nullptr, const i64* pre_s = nullptr, i64* victim_key = nullptr,
                                              i64* given_back = nullptr, int* n_given_back = nullptr) {
  u64 h;
  const u64 b0 = bucket0(key, v.nb, h);
  const u64 b1 = bucket1(h, b0, v.nb);
  claimed_empty = false;
  for (int attempt = 0; attempt < 256; ++attempt) {
    u64 best_score, best_word;
    i64 best_key;
    // key + score lines of both home buckets: four independent loads in flight
    i64 kk2[2], sc2[2];
    if (attempt == 0 && pre_k) {
      kk2[0] = pre_k[0]; kk2[1] = pre_k[1]; sc2[0] = pre_s[0]; sc2[1] = pre_s[1];
    } else {
      kk2[0] = load_key_coherent(key_line(v, b0) + sub); kk2[1] = load_key_coherent(key_line(v, b1) + sub);
      sc2[0] = (i64)__hip_atomic_load(score_line(v, b0) + sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sc2[1] = (i64)__hip_atomic_load(score_line(v, b1) + sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      keep_live(kk2[0], kk2[1], sc2[0], sc2[1]);
    }
    select_victim(b0, b1, kk2, sc2, sub, gshift, best_score, best_word, best_key);
    if (best_score == ~0ULL) continue;  // everything locked by concurrent evictors: look again
    if (best_key != EMPTY_KEY && !admit_always && in_score < best_score) return -1;
    i64 old = 0;
    if (sub == 0) old = (i64)atomicCAS((u64*)key_word(v, best_word), (u64)best_key, (u64)LOCKED_KEY);
    old = shfl_i64(old, gshift);
    if (old == best_key) {
      if (best_key != EMPTY_KEY) {
        // The key line and the score line were read by two unordered loads: a slot that a concurrent
        // evictor published in between shows its NEW key with the OLD (minimum) score of the entry it
        // replaced.  The slot is ours now, so its score is stable: re-read it and give the slot back
        // if it is not the score the choice was based on.
        unsigned lo = 0, hi = 0;
        if (sub == 0) {
          u64 now = __hip_atomic_load(score_word(v, best_word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          lo = (unsigned)now; hi = (unsigned)(now >> 32);
        }
        const u64 now = ((u64)(unsigned)__shfl((int)hi, gshift) << 32) | (unsigned)__shfl((int)lo, gshift);
        if (now != best_score) {
          if (sub == 0) __hip_atomic_store(key_word(v, best_word), best_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (the slot showed LOCKED for a moment: a lookup running beside this call may have missed the key)
          if (given_back && n_given_back) { if (*n_given_back < 4) given_back[*n_given_back] = best_key; *n_given_back += 1; }
          continue;
        }
      }
      claimed_empty = (best_key == EMPTY_KEY);
      *victim_word = best_word;
      if (victim_key) *victim_key = best_key;   // the key this call replaces (the overlapped step checks it against the next lookup)
      return (i64)((best_word >> 4) * SLOTS + (best_word & 15));
    }
  }
  return -3;
}

// The row and the score of a replaced slot must be in memory before the key replaces LOCKED_KEY (a concurrent
// evictor on another XCD that sees the new key must also see its new score).  An agent-scope release fence would
// do it, but on gfx950 that is an L2 write-back (buffer_wbl2) per call — 65 536 evictions of one batch spent
// ~5 ns each on it, 10x the rest of the kernel.  Instead the eviction path writes row and score WRITE-THROUGH
// (sc0 sc1: acknowledged by memory, nothing left dirty in this XCD's L2) and waits for those acknowledgements
// (vmcnt(0)) before the key store, which is itself an agent-scope atomic store.
__device__ __forceinline__ void publish_key(const TableView& v, u64 word, i64 key, int sub) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (sub == 0) __hip_atomic_store(key_word(v, word), key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// write-through stores for the eviction path (see publish_key)
__device__ __forceinline__ void store_wt8(void* p, u64 x) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(x) : "memory");
}
template <int G>
__device__ __forceinline__ void copy_bytes16_wt(unsigned char* dst, const unsigned char* src, unsigned bytes, int sub);

// Per-key score of the Hkv strategies (what eviction compares): LRU = device clock, LFU += in_score,
// EPOCH* = epoch << 32 | low word, CUSTOMIZED = caller's score (lookup_table_op_hkv.h:454-475).
template <bool WT = false>
__device__ __forceinline__ void update_score(const TableView& v, i64 row, bool is_new, int strategy,
                                             u64 in_score, u64 epoch, int sub) {
  if (!has_scores(v) || sub != 0 || row >= (i64)(v.nb * SLOTS)) return;
  u64 b;
  unsigned s;
  split_row((u64)row, b, s);
  u64* p = score_line(v, b) + s;
  u64 ns;
  switch (strategy) {
    case TFRA_EVICT_LFU: atomicAdd(p, in_score); return;  // slots are zeroed on clear/erase
    case TFRA_EVICT_EPOCHLRU: ns = (epoch << 32) | (wall_clock64() & 0xffffffffULL); break;
    case TFRA_EVICT_EPOCHLFU: {
      const u64 old = is_new ? 0 : *p;   // the only rule that reads the old score (a dependent access on a table of this size)
      u64 cnt = (old & 0xffffffffULL) + in_score;
      if (cnt > 0xffffffffULL) cnt = 0xffffffffULL;
      ns = (epoch << 32) | cnt;
    } break;
    case TFRA_EVICT_CUSTOMIZED: ns = in_score; break;
    default: ns = wall_clock64(); break;  // LRU: device-wide monotonic clock
  }
  if (WT) store_wt8(p, ns);
  else *p = ns;
}

__device__ __forceinline__ void size_add(const TableView& v, u64 wave_id, long long delta) {
  atomicAdd(&v.size_shards[(wave_id % SIZE_SHARDS) * SIZE_SHARD_STRIDE], (u64)delta);
}

// ---- row movement: G = copy granule in bytes (largest power of two <=16 dividing the row
// bytes and every base pointer); 16 lanes move 16*G bytes per step, fully coalesced -----------
template <int G> struct Granule;
template <> struct Granule<16> { typedef uint4 T; };
template <> struct Granule<8> { typedef uint2 T; };
template <> struct Granule<4> { typedef unsigned T; };
template <> struct Granule<2> { typedef unsigned short T; };
template <> struct Granule<1> { typedef unsigned char T; };

// ---- typed accumulate of one 16-B granule: r[j] = a[j] + b[j], ONE add per element (ValueArray::operator+=,
// K/lookup_impl/lookup_table_op_cpu.h:45-51); dt = tfra_dtype (uniform: a scalar branch) -----------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned add2_f16(unsigned a, unsigned b) {
  const _Float16 a0 = __builtin_bit_cast(_Float16, (unsigned short)a), a1 = __builtin_bit_cast(_Float16, (unsigned short)(a >> 16));
  const _Float16 b0 = __builtin_bit_cast(_Float16, (unsigned short)b), b1 = __builtin_bit_cast(_Float16, (unsigned short)(b >> 16));
  const _Float16 r0 = (_Float16)((float)a0 + (float)b0), r1 = (_Float16)((float)a1 + (float)b1);
  return (unsigned)__builtin_bit_cast(unsigned short, r0) | ((unsigned)__builtin_bit_cast(unsigned short, r1) << 16);
}
__device__ __forceinline__ unsigned add2_bf16(unsigned a, unsigned b) {
  const unsigned short r0 = f32_to_bf16(bf16_to_f32((unsigned short)a) + bf16_to_f32((unsigned short)b));
  const unsigned short r1 = f32_to_bf16(bf16_to_f32((unsigned short)(a >> 16)) + bf16_to_f32((unsigned short)(b >> 16)));
  return (unsigned)r0 | ((unsigned)r1 << 16);
}
__device__ __forceinline__ unsigned add4_i8(unsigned a, unsigned b) {   // four wrapping int8 adds
  return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
__device__ __forceinline__ uint4 add16_dt(uint4 a, uint4 b, int dt) {
  uint4 r;
  switch (dt) {
    case TFRA_F32:
      r.x = __float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x)); r.y = __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y));
      r.z = __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z)); r.w = __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w));
      break;
    case TFRA_F16: r.x = add2_f16(a.x, b.x); r.y = add2_f16(a.y, b.y); r.z = add2_f16(a.z, b.z); r.w = add2_f16(a.w, b.w); break;
    case TFRA_BF16: r.x = add2_bf16(a.x, b.x); r.y = add2_bf16(a.y, b.y); r.z = add2_bf16(a.z, b.z); r.w = add2_bf16(a.w, b.w); break;
    case TFRA_I8: r.x = add4_i8(a.x, b.x); r.y = add4_i8(a.y, b.y); r.z = add4_i8(a.z, b.z); r.w = add4_i8(a.w, b.w); break;
    case TFRA_I32: r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; r.w = a.w + b.w; break;
    case TFRA_I64: {
      const u64 s0 = (((u64)a.y << 32) | a.x) + (((u64)b.y << 32) | b.x), s1 = (((u64)a.w << 32) | a.z) + (((u64)b.w << 32) | b.z);
      r.x = (unsigned)s0; r.y = (unsigned)(s0 >> 32); r.z = (unsigned)s1; r.w = (unsigned)(s1 >> 32);
    } break;
    default: {   // TFRA_F64
      const double s0 = __longlong_as_double((i64)(((u64)a.y << 32) | a.x)) + __longlong_as_double((i64)(((u64)b.y << 32) | b.x));
      const double s1 = __longlong_as_double((i64)(((u64)a.w << 32) | a.z)) + __longlong_as_double((i64)(((u64)b.w << 32) | b.z));
      const u64 u0 = (u64)__double_as_longlong(s0), u1 = (u64)__double_as_longlong(s1);
      r.x = (unsigned)u0; r.y = (unsigned)(u0 >> 32); r.z = (unsigned)u1; r.w = (unsigned)(u1 >> 32);
    } break;
  }
  return r;
}

template <int G>
__device__ __forceinline__ void copy_bytes16(unsigned char* dst, const unsigned char* src,
                                             unsigned bytes, int sub) {
  typedef typename Granule<G>::T T;
  for (unsigned off = sub * G; off < bytes; off += 16 * G)
    *reinterpret_cast<T*>(dst + off) = *reinterpret_cast<const T*>(src + off);
}

// same, write-through for 16-B granules (eviction path, see publish_key)
template <int G>
__device__ __forceinline__ void copy_bytes16_wt(unsigned char* dst, const unsigned char* src, unsigned bytes, int sub) {
  typedef typename Granule<G>::T T;
  for (unsigned off = sub * G; off < bytes; off += 16 * G) {
    T t = *reinterpret_cast<const T*>(src + off);
    if (G == 16) store_wt16(dst + off, *reinterpret_cast<uint4*>(&t));
    else __hip_atomic_store(reinterpret_cast<T*>(dst + off), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- find: the work of ONE wave (16 keys: 4 per 16-lane group), shared by find_kernel (tfra_table.hip) and the launch that runs
// the lookup next to the de-duplication of the same ids (find_unique_kernel, tfra_csr.hip) -------------------------------------
// PF1: also put the SECOND home bucket's line of every key in flight with the first (a bounded table running
// near capacity: most b0 lines are full and flagged, ~1/3 of the resident keys and every miss need b1, and on
// a loaded memory round trip is ~2 us — two in flight beat two in a row).
template <int G, int U, bool WT, bool PF1>
__device__ __forceinline__ void find_wave(const TableView& v, size_t n, const i64* __restrict__ keys, unsigned char* __restrict__ out,
                                          uint8_t* __restrict__ exists, const unsigned char* __restrict__ defaults, int full,
                                          unsigned field_off, const unsigned wave) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const int grp = lane >> 4;
  constexpr int KPW = 4 * U;
  const unsigned base = wave * KPW;   // n < 2^32 (find_impl)
  if (base >= n) return;
  // Loads are kept UNCONDITIONAL (tail keys are clamped to the last valid index): a load inside an
  // `if (valid)` block gets its own `s_waitcnt vmcnt(0)` and the U probes / U rows would be fetched
  // one latency after the other instead of all in flight (measured 23 us -> 11 us per 131072 keys).
  const unsigned last = (unsigned)n - 1;
  // ONE LANE PER KEY for the scalar work: lane j (and j+16, j+32, j+48) holds key j of the wave's 16 and hashes it —
  // one instruction stream for 16 keys; the groups then fetch key / b0 / b1 of "their" key by shuffle.  (Hashing per
  // group cost 4x the instructions, and instruction issue, not HBM, is what bounds this kernel.)
  const i64 kreg = keys[min(base + (unsigned)(lane & (KPW - 1)), last)];
  u64 hreg;
  const unsigned b0reg = (unsigned)bucket0(kreg, v.nb, hreg);
  const unsigned b1reg = (unsigned)bucket1(hreg, b0reg, v.nb);
  i64 key[U];
  unsigned b0[U], b1[U], idx[U];
  i64 k0[U], k1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int j = u * 4 + grp;
    key[u] = shfl_i64(kreg, j);
    b0[u] = (unsigned)__shfl((int)b0reg, j);
    b1[u] = (unsigned)__shfl((int)b1reg, j);
    idx[u] = min(base + (unsigned)j, last);
    k0[u] = key_line(v, b0[u])[sub];  // U probes in flight
    k1[u] = PF1 ? key_line(v, b1[u])[sub] : 0;
  }
  static_assert(U == 4, "keep_live is written for U == 4");
  keep_live(k0[0], k0[1], k0[2], k0[3]);
  if (PF1) keep_live(k1[0], k1[1], k1[2], k1[3]);
  const unsigned char* src[U];
  unsigned char* dst[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const i64 word = probe_find_word(v, key[u], b0[u], b1[u], k0[u], sub, gshift, PF1 ? &k1[u] : nullptr);
    if (exists && sub == 0) exists[idx[u]] = word >= 0;
    src[u] = word >= 0 ? word_row_ptr(v, (u64)word) + field_off
                       : defaults + (full ? (u64)idx[u] * (u64)v.field_bytes : 0);
    dst[u] = out + (u64)idx[u] * (u64)v.field_bytes;
  }
  typedef typename Granule<G>::T T;
  for (unsigned off = sub * G; off < v.field_bytes; off += 16 * G) {
    T tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) tmp[u] = *reinterpret_cast<const T*>(src[u] + off);  // U rows in flight
    keep_live(tmp[0], tmp[1], tmp[2], tmp[3]);
#pragma unroll
    for (int u = 0; u < U; ++u) {  // clamped tail: same bytes twice
      if (G == 16 && WT) store_wt16(dst[u] + off, *reinterpret_cast<uint4*>(&tmp[u]));
      else *reinterpret_cast<T*>(dst[u] + off) = tmp[u];
    }
  }
}

}  // namespace tfra
