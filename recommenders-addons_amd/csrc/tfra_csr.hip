// Write-back of a batch whose ids MAY repeat, organised as CSR-by-key.
//
// Reference: the gradients of duplicate ids are summed, then ONE update per key
// (`_resource_apply_sparse_duplicate_indices` = unique + unsorted_segment_sum,
// PY/dynamic_embedding_optimizer.py:177-190) followed by (1+S) finds + dense apply + (1+S) upserts
// (:165-204); a plain upsert of a batch with repeats keeps the LAST occurrence
// (LaunchTensorsInsert on one thread, K/cuckoo_hashtable_op.cc:104-140).
//
// A Zipf-1.2 batch of 131 072 ids is extremely bimodal: ~790 keys occur more than 8 times and hold
// ~100 000 of the positions (the hottest ~24 000), the other ~21 000 keys share the remaining ~30 000.
//
//   PLAN (id-only, tfra_sparse_plan_build; any stream, typically one batch ahead on a second stream)
//     csr_tile_kernel    one block per 512 ids: equal ids grouped with an LDS hash; one descriptor
//                        (key, tile, count, where the tile keeps the positions) per distinct key of the tile,
//                        appended to the key's merge bucket (hash of the key).
//     csr_bucket_kernel  one block per bucket: groups its descriptors by key, orders a key's descriptors by tile
//                        (per-key tile table + wave scans), and emits
//                          cold keys (<= 8 occurrences): (key, first, count) + their batch positions, ascending;
//                          hot keys: their positions laid out in BINS of 512 entries — every 512 consecutive
//                            occurrences of a key fill one bin, the remainder (< 512, padded to 16) shares a bin
//                            with other remainders, never straddling — and (key, first partial, #partials).
//     csr_finish_kernel  publishes the counts, re-arms the cursors for the next build.
//   GRADIENT HALF (tfra_table_apply_planned)
//     hot_sums_kernel    one block per bin: 32 groups x 16 rows in flight, ordered add -> one partial row per run
//     apply_csr_kernel   one 16-lane group per unique key (hot keys first): gathers its <= 8 gradient rows or
//                        its partial rows IN ORDER, sums, locates/claims the table row, applies the optimizer.
//   ASSIGN (tfra_table_upsert_planned): upsert_own_kernel (one pass with bucket ownership) + upsert_rest_kernel (the few
//     keys that pass leaves over) copy the row of every key's LAST occurrence (no sums, any value dtype); the assign-only
//     plan (setplan_kernel, dim 0) is one kernel: distinct ids with their last position.
//
// Round 3, tried and dropped (measured on the 10^9-slot table, B = 131 072 Zipf-1.2; each is in the git history as one
// experiment, none in the code):
//   * the remainder pass deferred to the table's next call and taken into the lookup's kernel (its first 32 blocks run the
//     list, every block waits for them when the list is not empty): a lookup kernel that holds the remainder's code needs
//     81 registers instead of 57 (6 instead of 8 waves per SIMD, or spills on the remainder's path), an agent-scope acquire
//     per waiting block is an L2 invalidate each (60 us per launch), 2000 blocks polling one word take 300 us, and while the
//     blocks wait they hold every wave slot, so the plan kernel of the second stream cannot run beside them: 48-51 us per
//     step against 47; only a sequence whose lists are always empty gained (34 -> 28 us);
//   * keys that lost a claim handled inside the pass (found keys lock their own slot, evictions take their victim by
//     compare-and-swap): correct, 45 % of the lists become empty, but the remainder kernel costs the same 8 us with one item
//     as with sixteen, and the swaps cost the pass 1.5-3 us on batches that evict;
//   * the assign-only plan partitioned by KEY (every block reads all ids and keeps its hash range: no global hash table, no
//     device-scope atomics): one CU streams 1 MB of ids in >= 7 us — 73 us as written, ~13 us at best, against 16 us;
//   * the hot sums as the leading blocks of their consumer kernel (keys with partial rows wait for them): the bin part has
//     to acknowledge its write-through rows before it may count itself done and shares the consumer's registers — 22 us
//     instead of 12 inside a step, and the keys that need it cannot start before it ends: 37-45 us against 38.
//
// Summation tree of a key = f(its occurrence count) only: [16 consecutive occurrences, ascending batch position,
// sequential] -> [the 32 groups of a bin, sequential] -> [the key's partials, sequential]; keys with <= 8
// occurrences are summed strictly in batch order (bit-identical to the reference's sequential sum).  WHERE a
// key's records land (atomic cursors) differs from run to run, the values do not: results are bit-reproducible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/tfra_mi355x.h"
#include "tfra_device.h"
#include "tfra_host.h"
#include "tfra_optim_device.h"
#include "tfra_reduce_device.h"

using namespace tfra;
using namespace tfra::red;

namespace {

#ifndef TFRA_HOT_SUMS_HALVES
#define TFRA_HOT_SUMS_HALVES 1   // hot_sums_kernel: 8 rows in flight, twice (0: 16 at once, the form of rounds 2-5; A/B)
#endif
constexpr int DIRECT = 8;                 // occurrences the update kernel gathers by itself
constexpr int SEG = 512;                  // entries of a hot bin = rows one hot_sums block reduces
constexpr unsigned E_SKIP = 1u << 31, E_HEAD = 1u << 30, E_POS = (1u << 18) - 1;
constexpr unsigned TABW = 2048;           // u32 entries of the per-pass tile table (8 KB: 8 keys x 256 tiles per round)
constexpr unsigned CTR_STRIDE = 16;       // u64 words between two counters (one 128-B line each)
constexpr int MAXPASS_SPLIT = 64;
// Output space is handed out by ATOMIC counters, and same-line atomics of different workgroups serialise at ~25 ns
// each on this chip (1024 bucket blocks on ONE counter line: 25 us, measured).  So the outputs are split into NSH
// shards with private counters and private ranges: bucket b allocates in shard b % NSH (P/NSH = 16 atomics per line),
// and the last plan kernel publishes dense maps (keymap / binmap) over the shards for the kernels of the other half.
constexpr unsigned NSH = 64;
constexpr unsigned REC_WORDS = 16;        // a key record = 64 B: [0,1] key [2] count; few occurrences: [3] last position, [4..11] its batch
                                          // positions, ascending; many: [3] first partial row [4] #partials [5] entry
                                          // address of its last occurrence
constexpr unsigned KM_MANY = 1u << 31;    // keymap: the record lives in `hrec`

// global descriptor store: bucket b owns [b*CMAX, b*CMAX + cm); overflow list behind it
struct CsrDesc {
  i64* key;
  unsigned* ord;        // tile << 18 | (count == 1: position in tile, else: run index in the tile) << 9 | (count - 1)
  unsigned* cursor;     // [P] one per 128-B line
  i64* ovf_key;
  unsigned* ovf_ord;
  unsigned* ovf_bucket;
  unsigned* ovf_count;
  unsigned ovf_cap;
};

struct CsrOut {
  u64* counters;        // [NSH] one per line: few-keys | many-keys << 16 | partials << 32 | bins << 48; [NSH]: deferred keys
  unsigned* crec;       // [NSH*cr] records of the keys with <= DIRECT occurrences
  unsigned* hrec;       // [NSH*hr] records of the others
  unsigned* hent;       // [NSH*br] bins of SEG entries: batch position | E_HEAD | E_SKIP
  unsigned* hout;       // [NSH*br*32] per 16-entry item: partial row of the run starting there
  unsigned cr, hr, pr, br;   // per-shard capacities: records, records, partial rows, bins
};

// ---------------------------------------------------------------------------------------------
// plan kernel 1: per tile of 512 ids.
__global__ __launch_bounds__(NTA) void csr_tile_kernel(size_t n, const i64* __restrict__ ids, unsigned P, unsigned cm,
                                                       CsrDesc ds, unsigned* err,
                                                       unsigned* __restrict__ tile_entries,
                                                       unsigned short* __restrict__ run_start,
                                                       unsigned* __restrict__ tile_len, u64* counters) {
  // the allocation counters of this build start at zero (the bucket kernel, next on the stream, uses them)
  if (blockIdx.x == 0 && threadIdx.x <= NSH) counters[(size_t)threadIdx.x * CTR_STRIDE] = 0;
  constexpr unsigned GCAP = 2048;            // group table: 4x the tile => short probe chains
  constexpr int W = TILE / 32;               // words of a position bitmask
  constexpr int GM = GCAP / W;               // multi-member groups ranked per round (mask area = s_owner)
  __shared__ i64 s_key[TILE];
  __shared__ unsigned s_owner[GCAP];         // group table; afterwards the position masks
  __shared__ unsigned s_rep[GCAP];           // smallest input position of the group; later (m << 16 | base)
  __shared__ unsigned s_cnt[GCAP];
  __shared__ unsigned short s_list[TILE];    // positions of the multi-member groups, run after run, ascending
  __shared__ unsigned short s_run[TILE];     // ... and the run (multi-member group) each list entry belongs to
  __shared__ int s_scan[NTA / 64];
  unsigned* const s_mask = s_owner;
  const size_t tile = blockIdx.x, base = tile * TILE;
  const int nvalid = (int)min((size_t)TILE, n - base);
  const int p = threadIdx.x;
  const bool ok = p < nvalid;
  const i64 key = ok ? ids[base + p] : 0;
  s_key[p] = key;
  for (unsigned b = threadIdx.x; b < GCAP; b += NTA) { s_owner[b] = 0; s_rep[b] = 0xffffffffu; s_cnt[b] = 0; }
  __syncthreads();
  unsigned slot = 0;
  if (ok) {
    slot = lds_group_slot(s_key, s_owner, GCAP, p, fmix64((u64)key));
    atomicMin(&s_rep[slot], (unsigned)p);
    atomicAdd(&s_cnt[slot], 1u);
  }
  __syncthreads();
  const bool head = ok && s_rep[slot] == (unsigned)p;
  const unsigned cnt = ok ? s_cnt[slot] : 0;
  const bool mhead = head && cnt > 1;
  int n_multi, L;
  const int m = block_excl_scan<NTA>(mhead ? 1 : 0, s_scan, &n_multi);
  const int lbase = block_excl_scan<NTA>(mhead ? (int)cnt : 0, s_scan, &L);
  // descriptor append: the returning atomic is issued now and consumed after the ranking rounds
  unsigned desc_pos = 0, desc_bucket = 0;
  if (head) {
    desc_bucket = (unsigned)__umul64hi(fmix64((u64)key), (u64)P);
    desc_pos = atomicAdd(&ds.cursor[(size_t)desc_bucket * CSTRIDE], 1u);
  }
  if (mhead) s_rep[slot] = ((unsigned)m << 16) | (unsigned)lbase;
  __syncthreads();  // s_rep maps slot -> (m, base) for multi groups; s_owner is free
  for (int c0 = 0; c0 < n_multi; c0 += GM) {
    for (unsigned q = threadIdx.x; q < GCAP; q += NTA) s_mask[q] = 0;
    __syncthreads();
    const int mm = (ok && cnt > 1) ? (int)(s_rep[slot] >> 16) - c0 : -1;
    if (mm >= 0 && mm < GM) atomicOr(&s_mask[mm * W + (p >> 5)], 1u << (p & 31));
    __syncthreads();
    if (mm >= 0 && mm < GM) {
      int rank = __popc(s_mask[mm * W + (p >> 5)] & ((1u << (p & 31)) - 1u));
      for (int wds = 0; wds < (p >> 5); ++wds) rank += __popc(s_mask[mm * W + wds]);
      s_list[(int)(s_rep[slot] & 0xffffu) + rank] = (unsigned short)p;  // ascending input position inside the run
      s_run[(int)(s_rep[slot] & 0xffffu) + rank] = (unsigned short)(s_rep[slot] >> 16);
    }
    __syncthreads();
  }
  if (head) {
    const unsigned where = cnt == 1 ? (unsigned)p : (unsigned)m;
    if (cnt > 1) run_start[base + m] = (unsigned short)lbase;
    const unsigned ordv = ((unsigned)tile << 18) | (where << 9) | (cnt - 1);
    if (desc_pos < cm) {
      size_t d = (size_t)desc_bucket * CMAX + desc_pos;
      ds.key[d] = key; ds.ord[d] = ordv;
    } else {
      unsigned o = atomicAdd(ds.ovf_count, 1u);
      if (o < ds.ovf_cap) { ds.ovf_key[o] = key; ds.ovf_ord[o] = ordv; ds.ovf_bucket[o] = desc_bucket; }
      else atomicAdd(err, 1u);
    }
  }
  // the tile keeps its own member lists: csr_scatter_kernel moves them once the bucket kernel has said where to
  for (int q = threadIdx.x; q < L; q += NTA) tile_entries[base + q] = (unsigned)s_list[q] | ((unsigned)s_run[q] << 9);
  if (threadIdx.x == 0) tile_len[tile] = (unsigned)L;
}

// ---------------------------------------------------------------------------------------------
// plan kernel 2: per merge bucket.  Dynamic LDS: cm*36 + TABW*4 bytes.
__global__ __launch_bounds__(NT) void csr_bucket_kernel(unsigned P, unsigned ntiles, unsigned cm, CsrDesc ds,
                                                        uint4* __restrict__ drec, CsrOut out, unsigned* err) {
  extern __shared__ unsigned char smem[];
  i64* const e_key = reinterpret_cast<i64*>(smem);            // [cm] descriptors of the pass
  unsigned* const e_ord = reinterpret_cast<unsigned*>(e_key + cm);
  unsigned* const s_owner = e_ord + cm;                         // [cm] group table (1 + claiming descriptor)
  unsigned* const s_ct = s_owner + cm;                          // [cm] per group: #descriptors | entries << 11
  unsigned* const s_pref = s_ct + cm;                           // [cm] per descriptor: entries of its key in earlier tiles
  unsigned* const s_a = s_pref + cm;                            // [cm] per group: allocation, see below
  unsigned* const s_b = s_a + cm;
  unsigned* const s_tab = s_b + cm;                             // [TABW] tile table of the multi-descriptor keys
  unsigned short* const s_slot = reinterpret_cast<unsigned short*>(s_tab + TABW);  // [cm] group of each descriptor
  unsigned short* const s_hot = s_slot + cm;                    // [cm] head descriptors of the hot keys
  __shared__ int s_scan[NT / 64];
  __shared__ unsigned s_nhot, s_nshared, s_hot_ok;
  __shared__ unsigned s_cold_ok;
  __shared__ unsigned sh_ckbase, sh_hkbase, sh_pbase, sh_bbase;
  const unsigned b = blockIdx.x;
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = threadIdx.x >> 4, wv = threadIdx.x >> 6;
  const int n_b = (int)ds.cursor[(size_t)b * CSTRIDE];
  if (n_b == 0) return;
  const unsigned n_ovf = n_b > (int)cm ? min(*ds.ovf_count, ds.ovf_cap) : 0u;
  const int n_reg = min(n_b, (int)cm);
  // A bucket with more descriptors than one pass holds (several very hot keys hashing together) is processed in
  // npass = 2^k passes, pass q taking the keys with (hash >> 40) & (npass-1) == q.  Emission goes through atomic
  // cursors and cannot be redone, so npass is fixed up front from a histogram of the 64 finest classes.
  unsigned npass = 1;
  if (n_b > (int)cm) {
    __shared__ unsigned s_hist[MAXPASS_SPLIT];
    __shared__ unsigned s_np;
    if (threadIdx.x < MAXPASS_SPLIT) s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (int q = threadIdx.x; q < n_reg + (int)n_ovf; q += NT) {
      i64 k = 0; bool take = false;
      if (q < n_reg) { k = ds.key[(size_t)b * CMAX + q]; take = true; }
      else if (ds.ovf_bucket[q - n_reg] == b) { k = ds.ovf_key[q - n_reg]; take = true; }
      if (take) atomicAdd(&s_hist[(unsigned)(fmix64((u64)k) >> 40) & (MAXPASS_SPLIT - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned np = 2;
      for (; np < MAXPASS_SPLIT; np <<= 1) {
        unsigned worst = 0;
        for (unsigned c = 0; c < np; ++c) {
          unsigned sum = 0;
          for (unsigned j = c; j < MAXPASS_SPLIT; j += np) sum += s_hist[j];
          worst = max(worst, sum);
        }
        if (worst <= cm) break;
      }
      s_np = np;
    }
    __syncthreads();
    npass = s_np;
  }
  const unsigned ntile_pad = (ntiles + 63) & ~63u;         // table row length (<= 512)
  const int rows_per_round = (int)(TABW / ntile_pad);
  for (unsigned pass = 0; pass < npass; ++pass) {
    int n = 0;
    if (npass == 1) {
      for (int q = threadIdx.x; q < n_reg; q += NT) {
        size_t d = (size_t)b * CMAX + q;
        e_key[q] = ds.key[d]; e_ord[q] = ds.ord[d];
      }
      n = n_reg;
      __syncthreads();
    } else {  // filtered gather from the region and from the overflow list
      int carry = 0;
      bool too_many = false;
      for (int q0 = 0; q0 < n_reg + (int)n_ovf; q0 += NT) {
        int q = q0 + threadIdx.x;
        i64 k = 0; unsigned od = 0; bool take = false;
        if (q < n_reg) {
          size_t d = (size_t)b * CMAX + q;
          k = ds.key[d]; od = ds.ord[d]; take = true;
        } else if (q < n_reg + (int)n_ovf) {
          unsigned o = q - n_reg;
          if (ds.ovf_bucket[o] == b) { k = ds.ovf_key[o]; od = ds.ovf_ord[o]; take = true; }
        }
        take = take && (((unsigned)(fmix64((u64)k) >> 40) & (npass - 1)) == pass);
        int tot;
        int ex = block_excl_scan<NT>(take ? 1 : 0, s_scan, &tot);
        if (carry + tot > (int)cm) too_many = true;
        if (take && !too_many) { e_key[carry + ex] = k; e_ord[carry + ex] = od; }
        carry += tot;
      }
      __syncthreads();
      if (too_many) {  // even 64 classes are not enough (one key with > cm parts cannot happen: cm >= #tiles)
        if (threadIdx.x == 0) atomicAdd(err, 1u);
        continue;
      }
      n = carry;
    }
    if (n == 0) continue;
    // ---- group by key --------------------------------------------------------------------------
    for (unsigned q = threadIdx.x; q < cm; q += NT) { s_owner[q] = 0; s_ct[q] = 0; }
    if (threadIdx.x == 0) { s_nhot = 0; s_nshared = 0; s_hot_ok = 1; }
    __syncthreads();
    for (int q = threadIdx.x; q < n; q += NT) {
      unsigned slot = lds_group_slot(e_key, s_owner, cm, q, fmix64((u64)e_key[q]));
      atomicAdd(&s_ct[slot], 1u | (((e_ord[q] & 511u) + 1u) << 11));
      s_slot[q] = (unsigned short)slot;
      s_pref[q] = 0;
    }
    __syncthreads();
    // ---- entries of the same key in EARLIER tiles, for the keys with more than one descriptor: a row of
    //      per-tile counts per key, exclusive wave scan along the row -------------------------------
    int n_multi = 0;
    for (int pb = 0; pb < n; pb += NT) {
      int q = pb + threadIdx.x;
      unsigned slot = q < n ? s_slot[q] : 0;
      bool mh = q < n && s_owner[slot] - 1 == (unsigned)q && (s_ct[slot] & 2047u) > 1;
      int tm;
      int em = block_excl_scan<NT>(mh ? 1 : 0, s_scan, &tm);
      if (mh) s_a[slot] = (unsigned)(n_multi + em);
      n_multi += tm;
    }
    __syncthreads();
    for (int c0 = 0; c0 < n_multi; c0 += rows_per_round) {
      const int nrows = min(rows_per_round, n_multi - c0);
      for (unsigned q = threadIdx.x; q < (unsigned)nrows * ntile_pad; q += NT) s_tab[q] = 0;
      __syncthreads();
      for (int q = threadIdx.x; q < n; q += NT) {
        unsigned slot = s_slot[q];
        if ((s_ct[slot] & 2047u) > 1) {
          int mrow = (int)s_a[slot] - c0;
          if (mrow >= 0 && mrow < nrows) s_tab[mrow * ntile_pad + (e_ord[q] >> 18)] = (e_ord[q] & 511u) + 1u;
        }
      }
      __syncthreads();
      const unsigned epl = ntile_pad / 64;   // elements per lane (<= 8)
      for (int mrow = wv; mrow < nrows; mrow += NT / 64) {
        unsigned* row = s_tab + mrow * ntile_pad + lane * epl;
        unsigned loc = 0;
        for (unsigned e = 0; e < epl; ++e) loc += row[e];
        unsigned incl = loc;
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
          unsigned t2 = (unsigned)__shfl_up((int)incl, o2);
          if (lane >= o2) incl += t2;
        }
        unsigned run = incl - loc;
        for (unsigned e = 0; e < epl; ++e) { unsigned c = row[e]; row[e] = run; run += c; }
      }
      __syncthreads();
      for (int q = threadIdx.x; q < n; q += NT) {
        unsigned slot = s_slot[q];
        if ((s_ct[slot] & 2047u) > 1) {
          int mrow = (int)s_a[slot] - c0;
          if (mrow >= 0 && mrow < nrows) s_pref[q] = s_tab[mrow * ntile_pad + (e_ord[q] >> 18)];
        }
      }
      __syncthreads();
    }
    // ---- allocation.  keys with few occurrences: a block scan; the others (few keys): thread 0 ------------
    //   few : s_a = rank among the pass's such keys
    //   many: s_a = first partial (relative) | first full bin (relative) << 16, s_b = first 16-entry item of the remainder
    int nck = 0;
    for (int pb = 0; pb < n; pb += NT) {
      int q = pb + threadIdx.x;
      unsigned slot = q < n ? s_slot[q] : 0;
      bool hd = q < n && s_owner[slot] - 1 == (unsigned)q;
      unsigned tot = hd ? (s_ct[slot] >> 11) : 0;
      bool cold = hd && tot <= (unsigned)DIRECT;
      int tt;
      int ex = block_excl_scan<NT>(cold ? 1 : 0, s_scan, &tt);
      if (cold) s_a[slot] = (unsigned)(nck + ex);
      if (hd && !cold) { unsigned i = atomicAdd(&s_nhot, 1u); s_hot[i] = (unsigned short)q; }
      nck += tt;
    }
    __syncthreads();
    const unsigned shard = b & (NSH - 1);
    if (threadIdx.x == 0) {
      const unsigned nh = s_nhot;
      unsigned nshared = 0, used = 32, nfulltot = 0, npart = 0;
      for (unsigned i = 0; i < nh; ++i) {
        const unsigned slot = s_slot[s_hot[i]];
        const unsigned tot = s_ct[slot] >> 11, nfull = tot >> 9, rem = tot & 511u, it = (rem + 15) >> 4;
        unsigned remitem = 0;
        if (rem) {
          if (used + it > 32) { nshared++; used = 0; }
          remitem = (nshared - 1) * 32 + used;
          used += it;
        }
        s_a[slot] = npart | (nfulltot << 16);
        s_b[slot] = remitem;
        nfulltot += nfull;
        npart += nfull + (rem ? 1 : 0);
      }
      const unsigned nbins = nshared + nfulltot;
      const u64 c = atomicAdd(&out.counters[(size_t)shard * CTR_STRIDE],
                              (u64)nck | ((u64)nh << 16) | ((u64)npart << 32) | ((u64)nbins << 48));
      sh_ckbase = (unsigned)(c & 0xffffu); sh_hkbase = (unsigned)((c >> 16) & 0xffffu);
      sh_pbase = (unsigned)((c >> 32) & 0xffffu); sh_bbase = (unsigned)(c >> 48);
      s_nshared = nshared;
      // a shard's range is sized for twice its fair share (+ the single-key worst case for bins / partials): only a
      // pathological hash skew gets here; the keys are skipped and the build reports the error
      s_cold_ok = sh_ckbase + nck <= out.cr;
      s_hot_ok = sh_hkbase + nh <= out.hr && sh_pbase + npart <= out.pr && sh_bbase + nbins <= out.br;
      if (!s_cold_ok || (nh && !s_hot_ok)) atomicAdd(err, 1u);
    }
    __syncthreads();
    const unsigned nh = s_nhot, nshared = s_nshared;
    const bool hot_ok = s_hot_ok != 0, cold_ok = s_cold_ok != 0;
    const unsigned ckbase = shard * out.cr + sh_ckbase, hkbase = shard * out.hr + sh_hkbase,
                   pbase = shard * out.pr + sh_pbase, bbase = shard * out.br + sh_bbase;
    // shared bins start as "nothing here": every 16-entry item is its own empty run
    if (hot_ok)
      for (unsigned q = threadIdx.x; q < nshared * SEG; q += NT)
        out.hent[(size_t)bbase * SEG + q] = E_SKIP | ((q & 15u) == 0 ? E_HEAD : 0u);
    // key records
    if (cold_ok) {
      for (int q = threadIdx.x; q < n; q += NT) {
        unsigned slot = s_slot[q];
        if (s_owner[slot] - 1 != (unsigned)q) continue;
        unsigned tot = s_ct[slot] >> 11;
        if (tot <= (unsigned)DIRECT) {
          unsigned* rec = out.crec + (size_t)(ckbase + s_a[slot]) * REC_WORDS;
          rec[0] = (unsigned)(u64)e_key[q]; rec[1] = (unsigned)((u64)e_key[q] >> 32); rec[2] = tot;
        }
      }
    }
    if (hot_ok) {
      for (unsigned i = threadIdx.x; i < nh; i += NT) {
        const int q = s_hot[i];
        const unsigned slot = s_slot[q];
        const unsigned tot = s_ct[slot] >> 11, nfull = tot >> 9, rem = tot & 511u;
        const unsigned partrel = s_a[slot] & 0xffffu, fullrel = s_a[slot] >> 16, remitem = s_b[slot];
        const unsigned fullbin0 = bbase + nshared + fullrel;
        const unsigned last_addr = rem ? (bbase * SEG + remitem * 16 + rem - 1) : ((fullbin0 + nfull - 1) * SEG + SEG - 1);
        unsigned* rec = out.hrec + (size_t)(hkbase + i) * REC_WORDS;
        rec[0] = (unsigned)(u64)e_key[q]; rec[1] = (unsigned)((u64)e_key[q] >> 32); rec[2] = tot;
        rec[3] = pbase + partrel; rec[4] = nfull + (rem ? 1 : 0); rec[5] = last_addr;
        for (unsigned j = 0; j < nfull; ++j) out.hout[(size_t)(fullbin0 + j) * 32] = pbase + partrel + j;
        if (rem) out.hout[(size_t)bbase * 32 + remitem] = pbase + partrel + nfull;
      }
    }
    __syncthreads();   // the SKIP fill above is ordered before the entries below
    // ---- entries: descriptor (tile, where, c) owns entries [pref, pref + c) of its key ----------------
    auto write_entry = [&](unsigned slot, unsigned e, unsigned position) {
      const unsigned tot = s_ct[slot] >> 11;
      if (tot <= (unsigned)DIRECT) {
        if (cold_ok) {
          unsigned* rec = out.crec + (size_t)(ckbase + s_a[slot]) * REC_WORDS;
          rec[4 + e] = position;
          if (e + 1 == tot) rec[3] = position;   // the key's last occurrence (positions ascend with e)
        }
      } else if (hot_ok) {
        const unsigned nfull = tot >> 9, fullrel = s_a[slot] >> 16;
        size_t addr;
        bool hd;
        if (e < nfull * SEG) { addr = (size_t)(bbase + nshared + fullrel) * SEG + e; hd = (e & (SEG - 1)) == 0; }
        else { addr = (size_t)bbase * SEG + s_b[slot] * 16 + (e - nfull * SEG); hd = e == nfull * SEG; }
        out.hent[addr] = position | (hd ? E_HEAD : 0u);
      }
    };
    for (int q = threadIdx.x; q < n; q += NT) {
      const unsigned od = e_ord[q];
      if ((od & 511u) == 0) write_entry(s_slot[q], s_pref[q], ((od >> 18) << 9) | ((od >> 9) & 511u));
    }
    // descriptors with count >= 2: where their member list goes (csr_scatter_kernel does the moving, one thread per
    // member, all tiles at once — the block of the hottest bucket alone would take ~100 dependent round trips)
    for (int q = threadIdx.x; q < n; q += NT) {
      const unsigned od = e_ord[q];
      if ((od & 511u) == 0) continue;
      const unsigned slot = s_slot[q], tot = s_ct[slot] >> 11, e0 = s_pref[q];
      uint4 r;
      if (tot <= (unsigned)DIRECT) r = cold_ok ? make_uint4((ckbase + s_a[slot]) * REC_WORDS + 4 + e0, 0u, 0u, 0xffffffffu)
                                               : make_uint4(0u, 0u, 0u, 0xfffffffeu);
      else if (hot_ok) r = make_uint4(e0, (bbase + nshared + (s_a[slot] >> 16)) * SEG, bbase * SEG + s_b[slot] * 16, tot >> 9);
      else r = make_uint4(0u, 0u, 0u, 0xfffffffeu);   // dropped (capacity error reported)
      drec[(size_t)(od >> 18) * TILE + ((od >> 9) & 511u)] = r;
    }
    __syncthreads();
  }
}

// plan kernel 3 (the last one): per tile again.
//  (a) every member of a multi-member run goes to the place csr_bucket_kernel chose for its descriptor:
//      entry e = e0 + (rank inside the run) of its key;
//  (b) dense maps over the allocation shards: keymap[g] = record of the g-th unique key (keys with many occurrences
//      first), binmap[i] = i-th bin;
//  (c) block 0: publishes the counts (device copy for the kernels of the other half, pinned host copy for the
//      driver), re-arms the bucket cursors for the next build, latches the errors.
//   d_counts: [0] keys with many occurrences [1] other keys [2] - [3] bins [4] - [5] errors of the build
__global__ __launch_bounds__(NTA) void csr_scatter_kernel(const unsigned* __restrict__ tile_entries,
                                                          const unsigned short* __restrict__ run_start,
                                                          const unsigned* __restrict__ tile_len,
                                                          const uint4* __restrict__ drec, CsrOut out,
                                                          unsigned* __restrict__ keymap, i64* __restrict__ dkeys,
                                                          unsigned* __restrict__ binmap, unsigned* cursors, unsigned P,
                                                          unsigned* d_counts, unsigned* host_counts, unsigned gen) {
  __shared__ unsigned s_hpre[NSH + 1], s_cpre[NSH + 1], s_bpre[NSH + 1];
  if (threadIdx.x < 64) {   // wave 0: prefix sums of the shards' counts (clamped to the shard capacities)
    const unsigned sh = threadIdx.x;
    const u64 c = out.counters[(size_t)sh * CTR_STRIDE];
    const unsigned nc = min((unsigned)(c & 0xffffu), out.cr), nh = min((unsigned)((c >> 16) & 0xffffu), out.hr),
                   nb = min((unsigned)(c >> 48), out.br);
    unsigned ic = nc, ih = nh, ib = nb;
    for (int o2 = 1; o2 < 64; o2 <<= 1) {
      unsigned tc = (unsigned)__shfl_up((int)ic, o2), th = (unsigned)__shfl_up((int)ih, o2), tb = (unsigned)__shfl_up((int)ib, o2);
      if ((int)sh >= o2) { ic += tc; ih += th; ib += tb; }
    }
    s_cpre[sh + 1] = ic; s_hpre[sh + 1] = ih; s_bpre[sh + 1] = ib;
    if (sh == 0) { s_cpre[0] = 0; s_hpre[0] = 0; s_bpre[0] = 0; }
  }
  __syncthreads();
  const unsigned nhot = s_hpre[NSH], ncold = s_cpre[NSH], nbins = s_bpre[NSH];
  auto shard_of = [](const unsigned* pre, unsigned x) {   // largest s with pre[s] <= x
    unsigned lo = 0;
    for (unsigned step = NSH / 2; step > 0; step >>= 1) if (pre[lo + step] <= x) lo += step;
    return lo;
  };
  const unsigned gid = blockIdx.x * NTA + threadIdx.x;
  // dkeys[g] = the key itself: the write-back kernels start the table probe from it while keymap -> record resolves
  if (gid < nhot) {
    const unsigned sh = shard_of(s_hpre, gid), r = sh * out.hr + (gid - s_hpre[sh]);
    keymap[gid] = KM_MANY | r;
    dkeys[gid] = *reinterpret_cast<const i64*>(out.hrec + (size_t)r * REC_WORDS);
  } else if (gid < nhot + ncold) {
    const unsigned x = gid - nhot, sh = shard_of(s_cpre, x), r = sh * out.cr + (x - s_cpre[sh]);
    keymap[gid] = r;
    dkeys[gid] = *reinterpret_cast<const i64*>(out.crec + (size_t)r * REC_WORDS);
  }
  if (gid < nbins) {
    const unsigned sh = shard_of(s_bpre, gid);
    binmap[gid] = sh * out.br + (gid - s_bpre[sh]);
  }
  // (a)
  const size_t base = (size_t)blockIdx.x * TILE;
  const int L = (int)tile_len[blockIdx.x];
  const int q = threadIdx.x;
  if (q < L) {
    const unsigned te = tile_entries[base + q];
    const unsigned m = te >> 9, position = (unsigned)base + (te & 511u);
    const unsigned rank = (unsigned)q - run_start[base + m];
    const uint4 r = drec[base + m];
    if (r.w == 0xffffffffu) {
      const unsigned idx = r.x + rank, rb = idx & ~(REC_WORDS - 1);   // a record is 16 words, positions in words 4..11
      out.crec[idx] = position;
      if ((idx & (REC_WORDS - 1)) - 3 == out.crec[rb + 2]) out.crec[rb + 3] = position;   // the key's last occurrence
    } else if (r.w != 0xfffffffeu) {
      const unsigned e = r.x + rank, nfull = r.w;
      if (e < nfull * SEG) out.hent[(size_t)r.y + e] = position | ((e & (SEG - 1)) == 0 ? E_HEAD : 0u);
      else out.hent[(size_t)r.z + (e - nfull * SEG)] = position | (e == nfull * SEG ? E_HEAD : 0u);
    }
  }
  // (c) block 0: counts for the kernels of the other half (device copy) and for the host's grid sizes (pinned copy), cursors
  // re-armed for the next build, errors latched.  NOT a completion signal: other blocks are still scattering — whoever needs
  // the finished plan on another stream waits for the build's event (tfra_table_step_prefetch: hipEventQuery on the host), or
  // is ordered behind it by the stream.  (Round 2 treated the pinned generation as "plan complete" — ADVICE r2; a per-block
  // release + ticket here cost the kernel 8 us, the event costs nothing on the critical path.)
  if (blockIdx.x == 0) {
    __syncthreads();
    for (unsigned i = threadIdx.x; i <= P; i += NTA) cursors[(size_t)i * CSTRIDE] = 0;   // bucket cursors + overflow count
    if (threadIdx.x == 0) {
      const unsigned e = cursors[(size_t)(P + 1) * CSTRIDE];
      cursors[(size_t)(P + 1) * CSTRIDE] = 0;
      const unsigned v[6] = {nhot, ncold, 0u, nbins, 0u, e};
      for (int k = 0; k < 6; ++k) d_counts[k] = v[k];
      *reinterpret_cast<i64*>(d_counts + 32) = (i64)nhot + (i64)ncold;   // tfra_plan_partition: the count as tfra_partition reads it
      if (host_counts) {
        for (int k = 0; k < 6; ++k) __hip_atomic_store(host_counts + 1 + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_counts, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gradient half, kernel 1: one block per bin of 512 entries = 32 items of 16 entries, one 16-lane group per item.
// Runs are item-aligned (csr_bucket_kernel pads every run to whole items), so an item belongs to exactly one run or to
// none: the group loads its 16 entry words with one coalesced read, puts all 16 gradient rows in flight at once
// (unconditional loads, padding clamped to the item's first row), adds them in entry order, and the group holding the
// run's first item then adds the sums of the run's following items in item order (LDS) and writes the partial row.
template <int NCH>
__global__ __launch_bounds__(NTA) void hot_sums_kernel(const float* __restrict__ grads, int dim,
                                                       const unsigned* __restrict__ hent, const unsigned* __restrict__ hout,
                                                       const unsigned* __restrict__ binmap,
                                                       const unsigned* __restrict__ d_counts, float* __restrict__ partial,
                                                       unsigned* progress, unsigned progress_val) {
  constexpr int NG = NTA / 16;
  __shared__ float s_sum[NG][64];
  __shared__ unsigned char s_kind[NG + 1];   // 0 = item continues the run of the item before, 1 = first item of a run, 2 = empty item
  // tfra_table_step_prefetch: host-visible progress counter (pinned memory) — this kernel running means the
  // lookup of step `progress_val` and every earlier step of the main stream are complete
  if (progress && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(progress, progress_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, g = threadIdx.x >> 4;
  const unsigned nbins = d_counts[3];
  for (unsigned ib = blockIdx.x; ib < nbins; ib += gridDim.x) {
    const unsigned bin = binmap[ib];
    const unsigned e = hent[(size_t)bin * SEG + threadIdx.x];          // lane `sub` holds entry `sub` of the item
    const unsigned e0 = (unsigned)__shfl((int)e, gshift);
    const bool empty = (e0 & E_SKIP) != 0, first = (e0 & E_HEAD) != 0;
    const unsigned out_row = (first && !empty && sub == 0) ? hout[(size_t)bin * 32 + g] : 0u;
    const unsigned live = (unsigned)(__ballot(!(e & E_SKIP)) >> gshift) & 0xffffu;   // entries of the item that exist
    if (sub == 0) s_kind[g] = empty ? 2 : (first ? 1 : 0);
    if (threadIdx.x == 0) s_kind[NG] = 1;
    unsigned rows[16];   // element offset of each row (< 2^18 * 256)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const unsigned ej = (unsigned)__shfl((int)e, gshift + j);
      rows[j] = (((live >> j) & 1u) ? (ej & E_POS) : (e0 & E_POS)) * (unsigned)dim;
    }
    for (int k = 0; k < NCH; ++k) {
      const int col = k * 64 + sub * 4;
      const int cc = col < dim ? col : 0;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (TFRA_HOT_SUMS_HALVES && NCH == 1) {   // (rows of more than 64 floats keep 16 in flight: their kernels are at 122-128 registers either way)
      // 8 rows in flight, twice, instead of 16 at once: 76 registers instead of 106 => 6 waves per SIMD instead of 4 => the ~680 bins of
      // a Zipf batch (512-thread blocks) are resident in ONE round instead of two; the second batch of loads costs a trip, the second round
      // cost more: 10.0 -> 9.2 us under rocprofv3, configs[1]'s step 56.7-57.3 -> 55.6-55.7 us (A/B on one box, twice).  Same adds, same order.
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float4*>(grads + rows[h * 8 + j] + cc);
        keep_live(x[0], x[1], x[2], x[3]); keep_live(x[4], x[5], x[6], x[7]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if ((live >> (h * 8 + j)) & 1u) { acc.x += x[j].x; acc.y += x[j].y; acc.z += x[j].z; acc.w += x[j].w; }
      }
      } else {
      float4 x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = *reinterpret_cast<const float4*>(grads + rows[j] + cc);   // 16 rows in flight
      keep_live(x[0], x[1], x[2], x[3]); keep_live(x[4], x[5], x[6], x[7]);
      keep_live(x[8], x[9], x[10], x[11]); keep_live(x[12], x[13], x[14], x[15]);
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if ((live >> j) & 1u) { acc.x += x[j].x; acc.y += x[j].y; acc.z += x[j].z; acc.w += x[j].w; }
      }
      if (k) __syncthreads();   // the owners of the previous chunk have read s_sum
      *reinterpret_cast<float4*>(&s_sum[g][sub * 4]) = acc;
      __syncthreads();
      if (first && !empty) {
        for (int g2 = g + 1; s_kind[g2] == 0; ++g2) {
          const float4 y = *reinterpret_cast<const float4*>(&s_sum[g2][sub * 4]);
          acc.x += y.x; acc.y += y.y; acc.z += y.z; acc.w += y.w;
        }
        const unsigned orow = (unsigned)__shfl((int)out_row, gshift);
        if (col < dim) *reinterpret_cast<float4*>(partial + (size_t)orow * dim + col) = acc;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
struct CsrKeys {
  const unsigned* keymap;
  const i64* dkeys;
  const unsigned* crec; const unsigned* hrec;
  const unsigned* hent;
  const unsigned* d_counts;
  // SET plan (assign-only, see setplan_kernel): dense distinct keys, their slots, (last position + 1, occurrences) per slot
  const i64* ukeys;
  const unsigned* uslot;
  const struct SetEnt* sent;
};
// One slot of a SET plan's open-addressing table: 16 B, so that a probe is ONE load (the overlapped step's lookup probes the
// previous batch's plan for every id: find_fwd_role, tfra_step_impl.h)
struct SetEnt { i64 key; unsigned pos1, cnt; };   // key (EMPTY_KEY = free) | last position + 1 | occurrences
__device__ __forceinline__ uint2 set_pc(const SetEnt* e) { return *reinterpret_cast<const uint2*>(&e->pos1); }
// A SET plan's table as something to PROBE (read-only): is this key one of the batch's ids, and where is its last occurrence?
// Probe chains stay inside a WINDOW of SET_WIN consecutive slots (home slot = hash & (m2 - 1); the slot behind the window's last
// is its first): the overlapped step builds a plan one window per workgroup, in LDS, without atomics (tfra_step_impl.h), and
// every other builder and every prober follows the same rule.  (m2 >= 2 n slots for n ids: a window overflows never.)
constexpr unsigned SET_WIN = 2048, SET_WIN_LOG2 = 11;
constexpr unsigned SEG_CAP = 32;   // pairs per (window, tile) segment of a scatter (tfra_step_impl.h); more: the overflow list
__host__ __device__ __forceinline__ unsigned set_wmask(unsigned m2) { return (m2 < SET_WIN ? m2 : SET_WIN) - 1u; }
__device__ __forceinline__ unsigned set_at(unsigned slot, unsigned g, unsigned wm) { return (slot & ~wm) | ((slot + g) & wm); }   // g slots on, inside the window
struct SetProbe { const SetEnt* ent; unsigned m2; };   // m2 entries (a power of two) + the two sentinel slots + padding
__device__ __forceinline__ unsigned set_home(const SetProbe& p, i64 key, u64 h) {   // h = fmix64(key)
  return is_reserved_key(key) ? p.m2 + (unsigned)reserved_index(key) : (unsigned)(h >> 20) & (p.m2 - 1);
}
// All 16 lanes of a key group: does the table hold `key`?  Lanes 0..3 look at four consecutive entries per round (linear
// probing, slots are never freed during a build: a match anywhere is the key, an EMPTY entry before it ends the search).
__device__ __forceinline__ bool set_contains_group(const SetProbe& p, i64 key, int sub, int gshift) {
  const bool resv = is_reserved_key(key);
  unsigned slot = set_home(p, key, fmix64((u64)key));
  const unsigned wm = set_wmask(p.m2);
  for (int round = 0; round < 512; ++round) {
    const unsigned e = resv ? slot + (unsigned)(sub & 3) : set_at(slot, (unsigned)(sub & 3), wm);
    const i64 k = p.ent[e].key;
    const bool match = resv ? ((sub & 3) == 0 && k != EMPTY_KEY) : k == key;
    const unsigned mm = (unsigned)(__ballot(match && sub < 4) >> gshift) & 0xfu;
    const unsigned em = (unsigned)(__ballot(k == EMPTY_KEY && sub < 4) >> gshift) & 0xfu;
    if (mm) return true;
    if (em || resv) return false;
    slot = set_at(slot, 4u, wm);
  }
  return true;   // (a chain this long does not exist: 2 n slots for n ids; say "present", the conservative answer)
}

// The same for TWO plans at once (lanes 0..3 probe p, lanes 4..7 probe q: one round trip for both): is the key in either?
__device__ __forceinline__ bool set_contains_either_group(const SetProbe& p, const SetProbe& q, i64 key, int sub, int gshift) {
  const bool resv = is_reserved_key(key);
  const SetProbe& t = (sub & 4) ? q : p;
  unsigned slot = set_home(t, key, fmix64((u64)key));
  const unsigned wm = set_wmask(t.m2);
  unsigned open = 3u;   // bit 0: still looking in p, bit 1: in q
  for (int round = 0; round < 512 && open; ++round) {
    const unsigned e = resv ? slot + (unsigned)(sub & 3) : set_at(slot, (unsigned)(sub & 3), wm);
    const i64 k = t.ent[e].key;
    const bool match = resv ? ((sub & 3) == 0 && k != EMPTY_KEY) : k == key;
    const unsigned mm = (unsigned)(__ballot(match && sub < 8) >> gshift) & 0xffu;
    const unsigned em = (unsigned)(__ballot(k == EMPTY_KEY && sub < 8) >> gshift) & 0xffu;
    if ((mm & 0x0fu) && (open & 1u)) return true;
    if ((mm & 0xf0u) && (open & 2u)) return true;
    if ((em & 0x0fu) || resv) open &= ~1u;
    if ((em & 0xf0u) || resv) open &= ~2u;
    slot = set_at(slot, 4u, wm);
  }
  return open != 0;   // (a chain this long does not exist; "present" is the conservative answer)
}

// one coalesced 64-B load per key group: lane i holds word i of the key's record
__device__ __forceinline__ unsigned load_record(const CsrKeys& ks, unsigned g, int sub, bool& many) {
  const unsigned km = ks.keymap[g];
  many = (km & KM_MANY) != 0;
  return (many ? ks.hrec : ks.crec)[(size_t)(km & ~KM_MANY) * REC_WORDS + sub];
}

// The source rows of a key's sum, NB of them in flight: rows j0 .. j0+NB-1 of its list (clamped to the last one; loads issued
// together, adds in list order).  A key with few occurrences lists their batch positions in words 4.. of its record (lane i
// of the group holds word i: EVERY lane of the group must be here); a key with many lists consecutive rows of the partial
// sums.  The addresses are formed here, from the record word, not kept in an array across the kernel: with 8 pointers and
// 8 rows held per lane the update kernel needed 145 registers (3 waves per SIMD); this form needs 125 (Adam) / 109 (SGD).
template <int NB>
__device__ __forceinline__ void add_rows(float4& acc, const float* __restrict__ grads, const float* __restrict__ partial, bool hot,
                                         unsigned w, unsigned first, unsigned nsrc, unsigned j0, int dim, int c, int gshift) {
  float4 x[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const unsigned jj = min(j0 + (unsigned)j, nsrc - 1);
    const unsigned position = (unsigned)__shfl((int)w, gshift + 4 + (int)min(jj, 7u));
    const float* q = hot ? partial + (size_t)(first + jj) * dim : grads + (size_t)position * dim;
    x[j] = *reinterpret_cast<const float4*>(q + c);
  }
  if (NB == 4) keep_live(x[0], x[1], x[2], x[3]);
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (j0 + (unsigned)j < nsrc) { acc.x += x[j].x; acc.y += x[j].y; acc.z += x[j].z; acc.w += x[j].w; }
}

// the whole sum of a key; wmax = the largest list length (capped at 8) among the wave's four keys: the trip count of the
// common part is uniform across the wave, the few keys with more than 8 partial rows go on alone
__device__ __forceinline__ float4 sum_rows(const float* __restrict__ grads, const float* __restrict__ partial, bool hot, unsigned w,
                                           unsigned first, unsigned nsrc, unsigned wmax, int dim, int c, int gshift) {
  float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (wmax <= 1) add_rows<1>(gg, grads, partial, hot, w, first, nsrc, 0, dim, c, gshift);
  else if (wmax <= 2) add_rows<2>(gg, grads, partial, hot, w, first, nsrc, 0, dim, c, gshift);
  else {
    add_rows<4>(gg, grads, partial, hot, w, first, nsrc, 0, dim, c, gshift);
    if (wmax > 4) add_rows<4>(gg, grads, partial, hot, w, first, nsrc, 4, dim, c, gshift);
  }
  for (unsigned j0 = 8; j0 < nsrc; j0 += 4) add_rows<4>(gg, grads, partial, hot, w, first, nsrc, j0, dim, c, gshift);
  return gg;
}

// ---------------------------------------------------------------------------------------------
// gradient half, kernel 2: one 16-lane group per unique key, hot keys first (their partial lists are the longest
// chains of the kernel: started first, they finish inside the kernel's duration).
// PHASE2: bounded (Hkv) table at max_capacity — the keys flagged in `dflag` (no free slot in phase 1; one byte per key:
// a list appended through ONE atomic counter cost 4 ns per key, 260 us for a batch of new keys) replace the minimum-score
// entry of their two home buckets and start from the default row / initial slot values, exactly like
// apply_evict_kernel (tfra_optim.hip).
// (Tried: amdgpu_waves_per_eu(4) on the 145-register form — 128 VGPRs with 7 spilled: gradient half 32.5 us instead of 31.5.
// Without the pointer arrays — add_rows — it is 125 registers, 4 waves per SIMD, no spills: 28.7 us, step 57.2 instead of 59.9 us.
// Round 4: amdgpu_waves_per_eu(5, 5) on that form — 96 registers, 18 spilled for Adam: gradient half 28.7 -> 37.3 us, the step of
// configs[1] 55.4 -> 62.3 us (A/B on one box, twice).  Five waves need a kernel that NEEDS 96 registers, not one that spills to them.)
template <int KIND, bool PHASE2>
__global__ __launch_bounds__(256) void apply_csr_kernel(TableView v, OptP o, int dim, const float* __restrict__ grads,
                                                        const float* __restrict__ partial, CsrKeys ks,
                                                        const float* __restrict__ default_row, float aux0, float aux1,
                                                        ScoreP sp, uint8_t* __restrict__ dflag, unsigned* any_deferred,
                                                        unsigned use_gen) {
  if (PHASE2 && *any_deferred != use_gen) return;   // phase 1 of this use deferred nothing
  constexpr int S = NSlots<KIND>::v;
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const unsigned total = ks.d_counts[0] + ks.d_counts[1];
  const unsigned ngroups = (gridDim.x * blockDim.x) >> 4;
  int fresh = 0, failed = 0;
  if (o.d_lr) o.lr = *o.d_lr;
  if (!PHASE2 && blockIdx.x == 0 && threadIdx.x == 0 && ks.d_counts[5]) atomicAdd(v.err_count, ks.d_counts[5]);  // plan overflow
  // trips are uniform per wave (the batch width below is a wave-wide maximum): a group past the end re-reads the
  // last key's records and does nothing else
  for (unsigned wbase = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 2; wbase < total; wbase += ngroups) {
    const unsigned it_raw = wbase + (unsigned)(lane >> 4);
    const bool active = it_raw < total;
    const unsigned g = active ? it_raw : total - 1;
    if (PHASE2 && !__builtin_amdgcn_readfirstlane((int)(__ballot(active && dflag[g]) != 0))) continue;   // nothing deferred in this wave
    // two chains in flight: key -> first probe line, and keymap -> record -> source rows
    const i64 key = ks.dkeys[g];
    u64 h;
    const u64 b0 = bucket0(key, v.nb, h);
    i64 k0 = 0;
    if (!PHASE2) k0 = load_key_coherent(key_line(v, b0) + sub);
    bool hot;
    const unsigned w = load_record(ks, g, sub, hot);
    const unsigned cnt = (unsigned)__shfl((int)w, gshift + 2);
    const unsigned first = (unsigned)__shfl((int)w, gshift + 3);             // keys with many occurrences: first partial row
    const unsigned nsrc = hot ? (unsigned)__shfl((int)w, gshift + 4) : cnt;
    // wave-uniform batch width: 1 / 2 / 4 rows in flight (most keys of a Zipf batch occur once)
    unsigned wmax = min(nsrc, 8u);
    for (int o2 = 32; o2 >= 16; o2 >>= 1) wmax = max(wmax, (unsigned)__shfl_xor((int)wmax, o2));
    wmax = (unsigned)__builtin_amdgcn_readfirstlane((int)wmax);
    if (!active || (PHASE2 && !dflag[g])) continue;
    i64 row;
    bool is_new = false;
    u64 word = 0;
    bool claimed_empty = false;
    if (PHASE2) {
      const bool lru_like = sp.strategy == TFRA_EVICT_LRU || sp.strategy == TFRA_EVICT_EPOCHLRU;
      const u64 in_score = sp.strategy == TFRA_EVICT_EPOCHLFU ? ((sp.epoch << 32) | 1) : 1;
      row = evict_and_lock(v, key, in_score, lru_like, sub, gshift, &word, claimed_empty);
      is_new = true;
    } else {
      row = locate_or_claim_from(v, key, h, b0, k0, sub, gshift, is_new, sp.bounded);
      if (sp.bounded && sub == 0) {
        dflag[g] = row == NEED_EVICT;
        if (row == NEED_EVICT) *any_deferred = use_gen;
      }
    }
    if (row < 0) {
      failed += (sub == 0 && (PHASE2 ? row == -3 : row != NEED_EVICT));
      continue;
    }
    fresh += ((PHASE2 ? claimed_empty : is_new) && sub == 0);
    float* pr = reinterpret_cast<float*>(row_ptr(v, row));
    // (every lane of the group takes every trip — sum_rows reads the record words of the other lanes; a lane beyond the row
    // works on column 0 and stores nothing)
    for (int c0 = 0; c0 < dim; c0 += 64) {
      const bool col = c0 + sub * 4 < dim;
      const int c = col ? c0 + sub * 4 : 0;
      float4 p = *reinterpret_cast<const float4*>((is_new ? default_row : pr) + c);
      float4 s1 = *reinterpret_cast<const float4*>(pr + (S >= 1 ? dim : 0) + c);
      float4 s2 = *reinterpret_cast<const float4*>(pr + (S >= 2 ? 2 * dim : 0) + c);
      float4 gg = sum_rows(grads, partial, hot, w, first, nsrc, wmax, dim, c, gshift);
      float4 dummy = p;
      keep_live(dummy, p, s1, s2);
      if (is_new || S < 1) s1 = make_float4(aux0, aux0, aux0, aux0);
      if (is_new || S < 2) s2 = make_float4(aux1, aux1, aux1, aux1);
      apply_one<KIND>(o, gg.x, p.x, s1.x, s2.x);
      apply_one<KIND>(o, gg.y, p.y, s1.y, s2.y);
      apply_one<KIND>(o, gg.z, p.z, s1.z, s2.z);
      apply_one<KIND>(o, gg.w, p.w, s1.w, s2.w);
      // write-through: the rows leave L2 during the kernel, not at the boundary to the next one
      if (col) {
        store_wt16(pr + c, *reinterpret_cast<uint4*>(&p));
        if (S >= 1) store_wt16(pr + dim + c, *reinterpret_cast<uint4*>(&s1));
        if (S >= 2) store_wt16(pr + 2 * dim + c, *reinterpret_cast<uint4*>(&s2));
      }
    }
    // aux fields the optimizer does not own (table created with more slots than it uses)
    if (is_new && (int)v.n_fields - 1 > S) {
      for (int f = S + 1; f < (int)v.n_fields; ++f)
        for (int c = sub; c < dim; c += 16)
          __hip_atomic_store(pr + f * dim + c, (f == 1 ? aux0 : aux1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PHASE2) {
      if (sub == 0) store_wt8(score_word(v, word), 0);  // the slot starts a new life
      update_score<true>(v, row, true, sp.strategy, 1, sp.epoch, sub);
      publish_key(v, word, key, sub);
    } else {
      update_score(v, row, is_new, sp.strategy, 1, sp.epoch, sub);  // one write-back = one upsert
    }
  }
  for (int off = 32; off > 0; off >>= 1) { fresh += __shfl_xor(fresh, off); failed += __shfl_xor(failed, off); }
  if (lane == 0) {
    if (fresh) size_add(v, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, fresh);
    if (failed) atomicAdd(v.err_count, (unsigned)failed);
  }
}

// ---------------------------------------------------------------------------------------------
// tfra_reduce_by_key epilogue: the same per-key sums as apply_csr_kernel, written out instead of applied.
// dest != nullptr (tfra_plan_reduce_to): the sum of a key goes to row dest[p], p = the key's last batch position — the
// caller's map from batch positions to output rows (equal for all positions of a key), e.g. position -> owner-major
// index of the multi-GPU gradient route; keys_out / d_count are not written then.
__global__ __launch_bounds__(256) void gather_csr_kernel(int dim, const float* __restrict__ grads,
                                                         const float* __restrict__ partial, CsrKeys ks,
                                                         i64* __restrict__ keys_out, float* __restrict__ rows_out,
                                                         i64* __restrict__ d_count, const int* __restrict__ dest) {
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const unsigned total = ks.d_counts[0] + ks.d_counts[1];
  const unsigned ngroups = (gridDim.x * blockDim.x) >> 4;
  if (d_count && blockIdx.x == 0 && threadIdx.x == 0) *d_count = ks.d_counts[5] ? (i64)-1 : (i64)total;
  for (unsigned wbase = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 2; wbase < total; wbase += ngroups) {
    const unsigned it_raw = wbase + (unsigned)(lane >> 4);
    const bool active = it_raw < total;
    const unsigned g = active ? it_raw : total - 1;
    bool hot;
    const unsigned w = load_record(ks, g, sub, hot);
    const i64 key = (i64)(((u64)(unsigned)__shfl((int)w, gshift + 1) << 32) | (unsigned)__shfl((int)w, gshift));
    const unsigned cnt = (unsigned)__shfl((int)w, gshift + 2);
    const unsigned first = (unsigned)__shfl((int)w, gshift + 3);
    const unsigned nsrc = hot ? (unsigned)__shfl((int)w, gshift + 4) : cnt;
    unsigned wmax = min(nsrc, 8u);
    for (int o2 = 32; o2 >= 16; o2 >>= 1) wmax = max(wmax, (unsigned)__shfl_xor((int)wmax, o2));
    wmax = (unsigned)__builtin_amdgcn_readfirstlane((int)wmax);
    if (!active) continue;
    size_t orow = g;
    if (dest) {
      unsigned lastp = (unsigned)__shfl((int)w, gshift + (hot ? 5 : 3));   // few: the last position itself; many: where it is stored
      if (hot) lastp = ks.hent[lastp];
      orow = (size_t)dest[lastp & E_POS];
    }
    for (int c0 = 0; c0 < dim; c0 += 64) {   // (every lane takes every trip: see apply_csr_kernel)
      const bool col = c0 + sub * 4 < dim;
      const int c = col ? c0 + sub * 4 : 0;
      const float4 gg = sum_rows(grads, partial, hot, w, first, nsrc, wmax, dim, c, gshift);
      if (col) *reinterpret_cast<float4*>(rows_out + orow * dim + c) = gg;
    }
    if (keys_out && sub == 0) keys_out[g] = key;
  }
}

constexpr unsigned SLOW_CAP = 8192;   // items of the left-over list of an ownership pass (more: the flags of all keys are scanned)

// ---------------------------------------------------------------------------------------------
// ASSIGN write-back, single pass with BUCKET OWNERSHIP (without owner tags — TFRA_OPTION_NO_OWNER_TAGS — every key takes the
// locked protocol of upsert_rest_kernel).
// Every bucket has an owner tag (one 32-bit word in a dense side array — NOT in the bucket's key line: an atomic and a
// load issued together on the same 128-B line cost 41 us per 157 K instead of 11 us on separate lines,
// scripts/mb/atomic_probe.hip).  Every key of the launch swaps the launch's generation into the tags of its two home
// buckets (two atomic exchanges in flight with its line loads) and owns a bucket iff the tag it got back is from an
// older launch.  A key that owns BOTH home buckets is the
// only writer of the launch that can touch them — every other key of the launch whose sequence includes one of them
// fails that claim and leaves the table alone — so it resolves hit / free slot / minimum-score eviction with plain
// loads and stores: no CAS, no LOCKED state, no score re-read, no publish ordering, and ONE dependent round trip (the
// four lines and the two claims are in flight together) instead of the five of the locked protocol.
//
// LEFT-OVER keys: a key that loses a claim (two keys of one batch sharing a home bucket: ~(2U)^2 / (2 nb) of them, 16 of
// 23 K / 185 of 78 K on 10^9 slots) or that cannot be placed within its two home buckets appends a self-contained ITEM
// (key, value position, input score) to the launch's list; upsert_rest_kernel takes the items afterwards with the locked
// protocol.  A key that MAY live beyond its home buckets (both overflow flags set, ~0.1 % of the buckets of a table filled
// to capacity) is looked for with reads in the main pass and claims the bucket it is found in.
//
// Round 3 measurements on the 10^9-slot table (scripts/mb_own.py, mb_sweep.py; rocprofv3 per-kernel times):
//   * the pass costs 13 us + 0.23 us per 1000 keys: 3.0 us launch + key load, 5.5 us until the four lines AND the two
//     claims are back (the lines alone 3.5 us — every access is a TLB miss on 273 GB), 4 us of dependent ALU / cross-lane
//     work for ONE wave's 16 keys, 0.5 us value rows, 1 us stores;
//   * the claims are ~5 us of 32 (78 K keys), their footprint does not matter (tags folded into 8 MB: same time);
//   * tried and dropped: the left-over keys in two more OWNERSHIP rounds — by the last block of the pass (ticket) or by a
//     one-block kernel behind it — 14-19 us against 10-12 us for 32 blocks of the locked protocol (one workgroup is one
//     dependent chain per round, and the code of the rounds costs the pass registers); claim-after-look with shared /
//     exclusive claim words in the score lines (an atomic on a line that has just been read is still a fabric
//     read-modify-write, and it now sits behind the lines instead of beside them): 35 / 38 us against 30 / 24;
//     the value row prefetched with the lines (direct keys): 42 us against 31 (16 more registers per lane, spills).

// One left-over key with the locked protocol for every kind of write: locate or claim the key's slot, LOCK it (CAS key
// -> LOCKED: a concurrent evictor of this pass may have taken it, then start over), or lock a victim (evict_and_lock);
// write row and score write-through, publish the key.  With every writer of the pass holding its slot locked, an assign
// can no longer race with the eviction of the same slot, which is what the two separate kernels (assign / claim, then
// evict) are for when they handle a whole batch.
// hint (a left-over key of the ownership pass that FOUND its key but had lost a claim): the slot it saw the key in — locked
// straight away, without reading the lines again (two dependent round trips less for 15 of the 16 left-over keys of the
// metric's batch); somebody took the slot in between: the ordinary way.
template <int G>
__device__ __forceinline__ void locked_upsert_kv(const TableView& v, const unsigned char* __restrict__ vals, i64 key, unsigned last,
                                                 u64 in_score, const AuxInitPod& ai, const ScoreP& sp, int sub, int gshift,
                                                 int& fresh, int& failed, bool hinted = false, unsigned hint_word = 0,
                                                 i64* evicted_key = nullptr, int acc = 0, int acc_dt = 0, i64* given_back = nullptr,
                                                 int* n_given_back = nullptr) {
  // evicted_key (optional): set to the key this upsert replaced by eviction (untouched when it evicted nothing)
  // acc: the reference's insert_or_accum for this key (accumrase_fn, cuckoohash_map.hh:619-633) instead of an assign —
  //   1 (exists): present -> row += delta, one add per element; absent -> nothing.   2 (!exists): absent -> insert; present -> nothing
  const bool lru_like = sp.strategy == TFRA_EVICT_LRU || sp.strategy == TFRA_EVICT_EPOCHLRU;
  const u64 cmp = sp.strategy == TFRA_EVICT_EPOCHLFU ? ((sp.epoch << 32) | in_score) : in_score;
  i64 row = -1;
  u64 word = 0;
  bool is_new = false, evicted = false, side = false;
  if (hinted) {
    i64 old = 0;
    if (sub == 0) old = (i64)atomicCAS((u64*)key_word(v, (u64)hint_word), (u64)key, (u64)LOCKED_KEY);
    old = shfl_i64(old, gshift);
    if (old == key) { word = hint_word; row = (i64)((word >> 4) * SLOTS + (word & 15)); }
  }
  for (int attempt = 0; acc == 1 && attempt < 64 && row < 0; ++attempt) {   // accumulate: find the key (never claim a slot) and lock it
    const i64 r = probe_find<true>(v, key, sub, gshift);
    if (r < 0) return;                                   // absent & exists: dropped
    if (r >= (i64)(v.nb * SLOTS)) { row = r; side = true; break; }
    u64 rb;
    unsigned rs;
    split_row((u64)r, rb, rs);
    const u64 wd = rb * 16 + rs;
    i64 old = 0;
    if (sub == 0) old = (i64)atomicCAS((u64*)key_word(v, wd), (u64)key, (u64)LOCKED_KEY);
    old = shfl_i64(old, gshift);
    if (old == key) { row = r; word = wd; }              // else: an evictor of this pass took the slot; look again
  }
  for (int attempt = 0; acc != 1 && attempt < 64 && row < 0; ++attempt) {
    // This pass is one wave-lifetime of dependent round trips (~1.2 us each): both home buckets' key AND score lines
    // travel together up front (first attempt) instead of b0 -> b1 -> score lines one after the other.
    u64 h;
    const u64 b0 = bucket0(key, v.nb, h);
    const u64 b1 = bucket1(h, b0, v.nb);
    const bool pre = attempt == 0 && has_scores(v) && sp.bounded != 0;
    i64 kk2[2], sc2[2] = {0, 0};
    kk2[0] = load_key_coherent(key_line(v, b0) + sub);
    kk2[1] = pre ? load_key_coherent(key_line(v, b1) + sub) : 0;
    if (pre) {
      sc2[0] = (i64)__hip_atomic_load(score_line(v, b0) + sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sc2[1] = (i64)__hip_atomic_load(score_line(v, b1) + sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      keep_live(kk2[0], kk2[1], sc2[0], sc2[1]);
    }
    bool claimed = false;
    i64 r = locate_or_claim_from(v, key, h, b0, kk2[0], sub, gshift, claimed, sp.bounded, pre ? &kk2[1] : nullptr);
    if (r == NEED_EVICT) {
      bool ce = false;
      u64 wd = 0;
      i64 vk = EMPTY_KEY;
      r = evict_and_lock(v, key, cmp, lru_like, sub, gshift, &wd, ce, pre ? kk2 : nullptr, pre ? sc2 : nullptr, &vk, given_back, n_given_back);
      if (r == -1) break;                        // not admitted (its score is below every resident one): dropped
      if (r == -3) { failed += (sub == 0); break; }
      row = r; word = wd; is_new = true; evicted = !ce;
      if (evicted && evicted_key) *evicted_key = vk;
      fresh += (ce && sub == 0);
      break;
    }
    if (r < 0) { failed += (sub == 0); break; }
    if (acc == 2 && !claimed) return;                    // present & !exists: dropped (nothing was claimed or locked)
    fresh += (claimed && sub == 0);
    is_new = is_new || claimed;
    if (r >= (i64)(v.nb * SLOTS)) { row = r; side = true; break; }   // sentinel keys live in the side rows: nothing evicts there
    u64 rb;
    unsigned rs;
    split_row((u64)r, rb, rs);
    const u64 wd = rb * 16 + rs;
    i64 old = 0;
    if (sub == 0) old = (i64)atomicCAS((u64*)key_word(v, wd), (u64)key, (u64)LOCKED_KEY);
    old = shfl_i64(old, gshift);
    if (old == key) { row = r; word = wd; }   // else: an evictor of this pass took the slot; look again
  }
  if (row < 0) return;
  unsigned char* pr = row_ptr(v, row);
  if (acc == 1 && G == 16) {
    const unsigned char* dl = vals + (size_t)last * v.field_bytes;
    for (unsigned off = sub * 16; off < v.field_bytes; off += 256)
      store_wt16(pr + off, add16_dt(*reinterpret_cast<const uint4*>(pr + off), *reinterpret_cast<const uint4*>(dl + off), acc_dt));
  } else copy_bytes16_wt<G>(pr, vals + (size_t)last * v.field_bytes, v.field_bytes, sub);
  if (is_new) {
    for (unsigned f = 1; f < v.n_fields; ++f) {   // slot fields of the new row start at aux_init
      const unsigned pat = ai.pattern[(f - 1) & 3];
      unsigned char* q = pr + f * v.field_bytes;
      if ((v.field_bytes & 3) == 0)
        for (unsigned off = sub * 4; off < v.field_bytes; off += 64)
          __hip_atomic_store(reinterpret_cast<unsigned*>(q + off), pat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        for (unsigned off = sub; off < v.field_bytes; off += 16)
          __hip_atomic_store(q + off, (unsigned char)(pat >> (8 * (off % ai.elem_bytes))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (side) return;
  if (evicted && sub == 0) store_wt8(score_word(v, word), 0);   // the slot starts a new life
  update_score<true>(v, row, is_new, sp.strategy, in_score, sp.epoch, sub);
  publish_key(v, word, key, sub);
}

// the same for key g of a plan: key, last position and input score come from its record
template <int G>
__device__ __forceinline__ void locked_upsert_one(const TableView& v, const unsigned char* __restrict__ vals,
                                                  const u64* __restrict__ scores, const CsrKeys& ks, const AuxInitPod& ai,
                                                  const ScoreP& sp, unsigned g, int sub, int gshift, int& fresh, int& failed) {
  const i64 key = ks.dkeys[g];
  bool hot;
  const unsigned w = load_record(ks, g, sub, hot);
  const unsigned cnt = (unsigned)__shfl((int)w, gshift + 2);
  unsigned last = (unsigned)__shfl((int)w, gshift + (hot ? 5 : 3));
  if (hot) last = ks.hent[last];
  last &= E_POS;
  const u64 in_one = scores ? scores[last] : 1;
  const u64 in_score = sp.strategy == TFRA_EVICT_LFU ? (scores ? in_one : (u64)cnt) : in_one;
  locked_upsert_kv<G>(v, vals, key, last, in_score, ai, sp, sub, gshift, fresh, failed);
}

// A left-over key of the ownership pass, self-contained: the remainder pass needs nothing of the plan.
struct OwnItem {          // 32 B, written as two 16-B stores
  i64 key;
  unsigned last;          // batch position of the key's value row
  unsigned g;             // index of the key in the launch (its dflag byte)
  u64 ins;                // input score
  unsigned hinted, word;  // hinted != 0: the pass saw the key in slot `word` (bucket * 16 + slot) and did not write it
};
// Counters of one use by the ownership write-back (two sets alternate, the last kernel of use k zeroes the set of
// use k+1: nothing of use k-1 is still running by stream order).
struct OwnCtrs { unsigned n_a, spare[3]; };

// Where the keys of a launch come from:
//   SRC_PLAN    the unique keys of a de-duplication plan (value row = the key's LAST occurrence in the batch)
//   SRC_DIRECT  a caller's array of UNIQUE keys, value row i belongs to key i (tfra_table_insert_or_assign with
//               TFRA_FLAG_UNIQUE_KEYS: the reference's Insert op, hkv_hashtable_op_gpu.cu.cc:253-290)
//   SRC_SET     the distinct keys of a SET plan (assign-only: last position and count per key, no positions list)
enum { SRC_PLAN = 0, SRC_DIRECT = 1, SRC_SET = 2, SRC_GIVEN = 3 };   // SRC_GIVEN: the caller hands key and value position (the overlapped step: from LDS)

struct OwnArgs {
  TableView v;
  const unsigned char* vals;
  const u64* scores;
  CsrKeys ks;              // SRC_PLAN
  const i64* keys;         // SRC_DIRECT
  unsigned nkeys;          // SRC_DIRECT
  const long long* d_nkeys;   // SRC_DIRECT, optional: the key count on the device (nkeys = the buffers' length then)
  AuxInitPod ai;
  ScoreP sp;
  uint8_t* dflag;          // one byte per key of the launch: 4 = left over.  All zero between launches: only left-over keys
                           // are flagged, and whoever takes a left-over key clears its flag
  unsigned* tags;
  OwnItem* items;          // [item_cap] left-over list of the launch
  unsigned item_cap;
  const uint8_t* exists;   // ACC (insert_or_accum of unique keys, SRC_DIRECT): the caller's exists flag per key
  int acc_dt;              // ACC: tfra_dtype of the rows
  SetProbe own_set;        // HF outside the step launch (SRC_SET): the launch's own SET plan, as something to probe
  unsigned* stats_host;    // pinned (Table::own_stats_host) or null: where the remainder kernel leaves the pass's sample
};
// (OwnArgs stays a read-only kernel argument: a private, modified copy would live in scratch memory — its aux_init
// pattern is indexed dynamically — and every field access of the hot loop would become a scratch load.)
struct OwnFlags { bool with_scores, spec, lru, lru_like; };
__device__ __forceinline__ unsigned direct_count(const OwnArgs& a) {   // keys of a SRC_DIRECT launch
  if (!a.d_nkeys) return a.nkeys;
  const long long dn = *a.d_nkeys;
  return dn < 0 ? 0u : (unsigned)min((long long)a.nkeys, dn);
}

template <bool SIMPLE>
__device__ __forceinline__ OwnFlags own_setup(const OwnArgs& a) {
  OwnFlags fl;
  fl.with_scores = SIMPLE || has_scores(a.v);
  const bool dense = a.sp.bounded > 1 || (a.sp.bounded == 1 && *a.v.dense_flag);
  fl.spec = fl.with_scores && dense;   // an eviction is likely: the score lines travel with the key lines
  fl.lru = SIMPLE || a.sp.strategy == TFRA_EVICT_LRU;
  fl.lru_like = fl.lru || a.sp.strategy == TFRA_EVICT_EPOCHLRU;
  return fl;
}

// The keys the ownership pass leaves over: 32 blocks (a full grid on a small table, where they are most of the batch) of
// the locked protocol over the item list; the flags of ALL keys when the list overflowed.
template <int G, int SRC, bool ACC = false>
__global__ __launch_bounds__(256) void upsert_rest_kernel(const OwnArgs a, const unsigned* slow_ctr, unsigned* zero4) {
  // slow_ctr == nullptr: there was no ownership pass (no owner tags): EVERY key of the launch, with the locked protocol.
  // This kernel is a chain of dependent round trips for a handful of keys: the group's first item travels together with the
  // list's length (it is used only if the list turns out to reach that far), and the plan's key count is read only by the
  // launches that need it.
  const unsigned gi = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  uint4 f0 = make_uint4(0u, 0u, 0u, 0u), f1 = f0;
  if (slow_ctr) {
    const OwnItem* it = a.items + (gi < a.item_cap ? gi : 0u);
    f0 = reinterpret_cast<const uint4*>(it)[0];
    f1 = reinterpret_cast<const uint4*>(it)[1];
  }
  unsigned total = 0;
  if (!slow_ctr) total = SRC != SRC_DIRECT ? a.ks.d_counts[0] + a.ks.d_counts[1] : direct_count(a);
  const unsigned counted = slow_ctr ? *slow_ctr : total;
  if ((SRC == SRC_SET || SRC == SRC_DIRECT) && !ACC && slow_ctr && a.stats_host && blockIdx.x == 0 && threadIdx.x == 0) {   // the pass's sample -> the host (launch_own)
    __hip_atomic_store(a.stats_host, slow_ctr[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // not plain hits
    __hip_atomic_store(a.stats_host + 1, slow_ctr[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // keys looked at
  }
  if (zero4 && blockIdx.x == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0;   // last kernel of this use: arm the next use's counters
  if (counted == 0) return;
  const bool listed = slow_ctr && counted <= a.item_cap;
  if (slow_ctr && !listed) total = SRC != SRC_DIRECT ? a.ks.d_counts[0] + a.ks.d_counts[1] : direct_count(a);
  const unsigned n = listed ? counted : total;
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48;
  const unsigned ngroups = (gridDim.x * blockDim.x) >> 4;
  int fresh = 0, failed = 0;
  for (unsigned i = gi; i < n; i += ngroups) {
    if (listed) {
      uint4 w0 = f0, w1 = f1;
      if (i != gi) {
        w0 = reinterpret_cast<const uint4*>(a.items + i)[0];
        w1 = reinterpret_cast<const uint4*>(a.items + i)[1];
      }
      const i64 key = (i64)(((u64)w0.y << 32) | w0.x);
      locked_upsert_kv<G>(a.v, a.vals, key, w0.z, ((u64)w1.y << 32) | w1.x, a.ai, a.sp, sub, gshift, fresh, failed, (w1.z & 1u) != 0, w1.w, nullptr,
                          ACC ? ((w1.z & 2u) ? 1 : 2) : 0, a.acc_dt);
      if (sub == 0) a.dflag[w0.w] = 0;
    } else {
      if (slow_ctr && a.dflag[i] != 4) continue;
      if (SRC == SRC_PLAN) locked_upsert_one<G>(a.v, a.vals, a.scores, a.ks, a.ai, a.sp, i, sub, gshift, fresh, failed);
      else if (SRC == SRC_SET) {
        const uint2 pc = set_pc(a.ks.sent + a.ks.uslot[i]);
        const u64 in_one = a.scores ? a.scores[pc.x - 1] : 1;
        locked_upsert_kv<G>(a.v, a.vals, a.ks.ukeys[i], pc.x - 1, a.sp.strategy == TFRA_EVICT_LFU ? (a.scores ? in_one : (u64)pc.y) : in_one,
                            a.ai, a.sp, sub, gshift, fresh, failed);
      } else locked_upsert_kv<G>(a.v, a.vals, a.keys[i], i, a.scores ? a.scores[i] : 1, a.ai, a.sp, sub, gshift, fresh, failed, false, 0, nullptr,
                                 ACC ? (a.exists[i] ? 1 : 2) : 0, a.acc_dt);
      if (slow_ctr && sub == 0) a.dflag[i] = 0;
    }
  }
  for (int off = 32; off > 0; off >>= 1) { fresh += __shfl_xor(fresh, off); failed += __shfl_xor(failed, off); }
  if (lane == 0) {
    if (fresh) size_add(a.v, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, fresh);
    if (failed) atomicAdd(a.v.err_count, (unsigned)failed);
  }
}

// 16-lane minimum of (score, index) pairs with DPP row rotations (row_ror 8, 4, 2, 1: every lane ends up with the row's
// minimum; ~8 cycles per move instead of an LDS round trip per __shfl_xor — the four dependent rounds of the victim choice
// were a fifth of the 4 us a wave spends deciding)
__device__ __forceinline__ unsigned dpp_ror(unsigned x, int n) {
  switch (n) {
    case 8: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false);
    case 4: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, false);
    case 2: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xf, 0xf, false);
    default: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xf, 0xf, false);
  }
}
// Victim choice among the 30 slots of (b0, b1), as select_victim_merged (tfra_device.h): minimum score, the lower index
// (b0's slots before b1's) on ties, EMPTY counts as score 0, LOCKED slots are not candidates.
__device__ __forceinline__ void select_victim_dpp(u64 b0, u64 b1, const i64 (&kk2)[2], const i64 (&sc2)[2], int sub, int gshift,
                                                  u64& best_score, u64& best_word) {
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
    const u64 os = ((u64)dpp_ror((unsigned)(my >> 32), o) << 32) | dpp_ror((unsigned)my, o);
    const unsigned oi = dpp_ror(idx, o);
    if (os < my || (os == my && oi < idx)) { my = os; idx = oi; }
  }
  best_score = my;
  best_word = (idx >= 16u ? b1 : b0) * 16 + (idx & 15u);
}

// keep_live over U in-flight values (U = 2 or 4)
template <int U, typename T>
__device__ __forceinline__ void keep_live_u(T (&x)[U]) {
  if (U == 4) keep_live(x[0], x[1], x[2], x[3]);
  else keep_live(x[0], x[U - 1], x[0], x[U - 1]);
}
template <int U, typename T>
__device__ __forceinline__ void keep_live_u2(T (&x)[U][2], int k) {
  if (U == 4) keep_live(x[0][k], x[1][k], x[2][k], x[3][k]);
  else keep_live(x[0][k], x[U - 1][k], x[0][k], x[U - 1][k]);
}

// One batch of 16 keys of one wave.  What is scalar per key — the key word, its hash, the two ownership claims, the plan
// record (count, last position), the input score — is done ONE LANE PER KEY (lane j of every group holds key j; group 0
// issues the claims): one instruction stream for 16 keys.  What needs a whole line — the four bucket lines, the ballots,
// the victim choice, the row copy — is done one 16-lane group per key, 4 keys per group in flight (the scalar results
// reach the group by shuffle).  SIMPLE: the common shape — rows without optimizer slots, LRU scores, no caller scores —
// with everything else compiled out.
// ORDER (round 3, from the phase timings above): the clock is read first (s_memrealtime is slow); every cross-lane
// broadcast is issued before the first use of any; the claims are issued BEHIND the line loads and consumed LAST — memory
// returns in order, so a claim issued first would hold back the lines, and consumed first it would stall the decisions,
// which do not need it.
// gj = the lane's key: index into the plan's dense keys / the caller's key array (clamped to a valid index; `valid` says
// whether the lane's key is real).
// U: keys per 16-lane group in flight (the wave's batch is 4 U keys: lanes 0 .. 4U-1 of every group hold them).  2 for up
// to a batch's worth of keys (22.7 K keys are 1420 waves of 16 keys on 1024 SIMDs: the 4 us of dependent cross-lane work
// per wave halve, the waves double and still fit in one round), 4 beyond.
// CF (the overlapped step, tfra_step_impl.h): the lookup of the NEXT batch runs beside this pass and reads the table rows of every key
// that is not in this batch — an entry this pass is about to EVICT must not be one of them: `cf` = the next batch's plan; a victim
// that is in it defers the new key to the remainder pass (which runs after that lookup and corrects its output).
// ACC (SRC_DIRECT, 16-B granules): the reference's insert_or_accum of unique keys (accumrase_fn, cuckoohash_map.hh:619-633;
// HkvHashTableOfTensorsGpu::Accum, K/hkv_hashtable_op_gpu.cu.cc:292-335) instead of an assign — a.exists[i] set: the key is
// expected in the table, present => row += value row (one add per element), absent => dropped; not set: absent => insert,
// present => dropped.  A dropped key writes nothing, whatever its claims say (the keys of a call are unique: any order of
// them is a valid serial order).
template <int G, bool SIMPLE, int SRC, int U, bool CF = false, bool ACC = false, bool HF = false>
__device__ __forceinline__ void own_batch16(const OwnArgs& a, const OwnFlags fl, unsigned gj, bool valid, unsigned gen, unsigned* slow_ctr,
                                            int lane, int& fresh, const SetProbe* cf = nullptr, unsigned* cf_stat = nullptr,
                                            i64 kgiven = 0, unsigned lastgiven = 0, const SetProbe* own_plan = nullptr, int* not_hits = nullptr) {
  constexpr bool hf = HF;   // (CF && HF: the overlapped step's launch; HF alone: upsert_own_kernel over a SET plan, victims checked against that plan)
  const u64* const scores = SIMPLE ? nullptr : a.scores;
  const TableView& v = a.v;
  const CsrKeys& ks = a.ks;
  const int sub = lane & 15, gshift = lane & 48, grp = lane >> 4;
  const u64 now = fl.lru_like ? (u64)wall_clock64() : 0;   // one clock read for the 16 keys (LRU scores tie within a wave)
  // ---- one lane per key ------------------------------------------------------------------------------------
  i64 kreg;
  unsigned kmreg = 0, lastreg = gj;
  u64 insreg = 1;
  if (SRC == SRC_PLAN) { kreg = ks.dkeys[gj]; kmreg = ks.keymap[gj]; }
  else if (SRC == SRC_SET) { kreg = ks.ukeys[gj]; kmreg = ks.uslot[gj]; }
  else if (SRC == SRC_GIVEN) { kreg = kgiven; lastreg = lastgiven; }
  else kreg = a.keys[gj];
  const unsigned exreg = ACC ? (unsigned)a.exists[gj] : 0u;
  u64 hreg;
  const unsigned b0reg = (unsigned)bucket0(kreg, v.nb, hreg);
  const unsigned b1reg = (unsigned)bucket1(hreg, b0reg, v.nb);
  const bool reserved = is_reserved_key(kreg);   // sentinel keys live in the side rows: the general path
  // ---- one group per key: every broadcast first, then the lines of 4 keys in flight -------------------------
  i64 key[U], kk[U][2], sc[U][2];
  unsigned b0[U], b1[U], gk[U];
  bool on[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int j = u * 4 + grp;
    key[u] = shfl_i64(kreg, j);
    b0[u] = (unsigned)__shfl((int)b0reg, j);
    b1[u] = (unsigned)__shfl((int)b1reg, j);
    gk[u] = (unsigned)__shfl((int)gj, j);
    on[u] = __shfl((int)(valid && !reserved), j) != 0;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // plain loads: everything written before this launch is visible, and nobody else writes a bucket this key owns
    kk[u][0] = key_line(v, b0[u])[sub];
    kk[u][1] = key_line(v, b1[u])[sub];
    // (HF, the step launch: 96 % of the keys are hits, which write their score word and never read a score line — the lines are fetched
    // below, and only by the waves that hold a key in need of a victim: 5.8 MB less random reads per launch on the metric's stream)
    sc[u][0] = (fl.spec && !hf) ? (i64)score_line(v, b0[u])[sub] : 0;
    sc[u][1] = (fl.spec && !hf) ? (i64)score_line(v, b1[u])[sub] : 0;
  }
  // ---- per key again, while the lines travel: count and last position from the plan record, input score ---
  if (SRC == SRC_PLAN) {
    const bool hot = (kmreg & KM_MANY) != 0;
    const unsigned* rec = (hot ? ks.hrec : ks.crec) + (size_t)(kmreg & ~KM_MANY) * REC_WORDS;
    const uint2 cl = *reinterpret_cast<const uint2*>(rec + 2);   // (count, last position of a key with few occurrences)
    const unsigned cnt = cl.x;
    lastreg = cl.y;
    if (hot) lastreg = ks.hent[rec[5]];                          // many: where it is stored
    lastreg &= E_POS;
    const u64 in_one = scores ? scores[lastreg] : 1;
    insreg = a.sp.strategy == TFRA_EVICT_LFU ? (scores ? in_one : (u64)cnt) : in_one;
  } else if (SRC == SRC_SET) {
    const uint2 pc = set_pc(ks.sent + kmreg);   // (last position + 1, occurrences) of the key's slot in the plan's table
    lastreg = pc.x - 1;
    const u64 in_one = scores ? scores[lastreg] : 1;
    insreg = a.sp.strategy == TFRA_EVICT_LFU ? (scores ? in_one : (u64)pc.y) : in_one;
  } else if (SRC != SRC_GIVEN) {
    insreg = scores ? scores[lastreg] : 1;
  }
  // the claims, behind the loads in program order (a clamped duplicate must not claim: it would lock out the real key)
  unsigned c0 = 0, c1 = 0;
  if (!hf && grp == 0 && valid && !reserved) {
    c0 = atomicExch(a.tags + b0reg, gen);
    c1 = atomicExch(a.tags + b1reg, gen);
  }
  keep_live_u2<U>(kk, 0);
  keep_live_u2<U>(kk, 1);
  if (fl.spec && !hf) {
    keep_live_u2<U>(sc, 0);
    keep_live_u2<U>(sc, 1);
  }
  // ---- what each key would do, from its lines alone (the claims are still travelling) -----------------------
  u64 word[U], in_s[U];
  int act[U];   // 0 nothing to write, 1 assign (hit), 2 new key in a free slot, 3 new key over an evicted entry
  int why[U];   // 0 handled, 1 lost a claim, 2 cannot be placed within the home buckets: the locked protocol
  bool flag_b0[U];   // the key goes to b1 although b0 never overflowed before: finds must go on to b1
  unsigned bxc[U];   // bucket beyond the home buckets the key was found in (it must be claimed too); ~0: none
  bool ex[U];        // ACC: the caller's exists flag
  bool need_v[U], ovf0_u[U];   // HF: the key needs a victim (score lines fetched below); b0's overflow flag as the decision saw it
  auto choose_victim = [&](int u, bool ovf0) {
    u64 best_score, best_word;
    select_victim_dpp(b0[u], b1[u], kk[u], sc[u], sub, gshift, best_score, best_word);
    const u64 cmp = a.sp.strategy == TFRA_EVICT_EPOCHLFU ? ((a.sp.epoch << 32) | in_s[u]) : in_s[u];
    if (fl.lru_like || cmp >= best_score) {   // else: not admitted, dropped like HKV does
      word[u] = best_word;
      act[u] = 3;
      flag_b0[u] = !ovf0 && (best_word >> 4) == b1[u];
      if (CF || HF) {
        const int vsrc = gshift + (int)(best_word & 15u);
        const i64 ka = shfl_i64(kk[u][0], vsrc), kb = shfl_i64(kk[u][1], vsrc);
        const i64 vk = (best_word >> 4) == (u64)b1[u] ? kb : ka;
        // CF: the next lookup wants it; HF: nor may it be a key of THIS batch — those are written without a claim, see below
        bool wanted = false;
        if (vk != EMPTY_KEY) {
          if (CF && HF) wanted = set_contains_either_group(*cf, *own_plan, vk, sub, gshift);
          else if (CF) wanted = set_contains_group(*cf, vk, sub, gshift);
          else wanted = own_plan ? set_contains_group(*own_plan, vk, sub, gshift) : true;   // (no plan to ask: defer — launch_own never picks HF then)
        }
        if (wanted) {   // deferred to the remainder
          act[u] = 0; why[u] = 3;
          if (cf_stat && sub == 0) atomicAdd(cf_stat, 1u);
        }
      }
    }
  };
#pragma unroll
  for (int u = 0; u < U; ++u) {
    need_v[u] = false; ovf0_u[u] = false;
    ex[u] = ACC && __shfl((int)exreg, u * 4 + grp) != 0;
    in_s[u] = 1;
    if (!fl.lru_like) in_s[u] = (u64)shfl_i64((i64)insreg, u * 4 + grp);   // (LRU-type scores ignore the input score)
    act[u] = 0; why[u] = 0; word[u] = 0; flag_b0[u] = false; bxc[u] = ~0u;
    if (!on[u]) continue;
    const unsigned hit0 = (unsigned)(__ballot(sub < SLOTS && kk[u][0] == key[u]) >> gshift) & 0x7fffu;
    const unsigned hit1 = (unsigned)(__ballot(sub < SLOTS && kk[u][1] == key[u]) >> gshift) & 0x7fffu;
    const unsigned emp0 = (unsigned)(__ballot(sub < SLOTS && kk[u][0] == EMPTY_KEY) >> gshift) & 0x7fffu;
    const unsigned emp1 = (unsigned)(__ballot(sub < SLOTS && kk[u][1] == EMPTY_KEY) >> gshift) & 0x7fffu;
    const bool ovf0 = ((__ballot(sub == 15 && ((u64)kk[u][0] & META_OVF0)) >> gshift) & 0xffffu) != 0;
    const bool ovf1 = ((__ballot(sub == 15 && ((u64)kk[u][1] & META_OVF1)) >> gshift) & 0xffffu) != 0;
    if (hit0) { word[u] = (u64)b0[u] * 16 + (__ffs(hit0) - 1); act[u] = 1; }
    else if (hit1) { word[u] = (u64)b1[u] * 16 + (__ffs(hit1) - 1); act[u] = 1; }
    else {
      bool absent = !(ovf0 && ovf1);   // the flags end the search at b0 / b1
      if (!absent) {
        // The key may live further along (placed while the table still walked): follow the flags with READS.  Found in
        // bucket bx: it claims bx too — every key that could evict from bx has bx as a home bucket and claimed it at its
        // start, so the exchange tells who goes first.
        unsigned bx = b1[u];
#pragma unroll 1
        for (int stepn = 0; stepn < 8 && !absent && !act[u] && !why[u]; ++stepn) {
          bx = bx + 1 == (unsigned)v.nb ? 0u : bx + 1;
          const i64 kx = load_key_coherent(key_line(v, bx) + sub);
          const unsigned hitx = (unsigned)(__ballot(sub < SLOTS && kx == key[u]) >> gshift) & 0x7fffu;
          if (hitx) { word[u] = (u64)bx * 16 + (__ffs(hitx) - 1); act[u] = 1; if (bx != b0[u] && bx != b1[u]) bxc[u] = bx; }
          else if (!((__ballot(sub == 15 && ((u64)kx & META_OVF1)) >> gshift) & 0xffffu)) absent = true;
          else if (stepn == 7) why[u] = 2;   // a long chain (an unbounded table): the general path
        }
      }
      if (absent && !(ACC && ex[u])) {   // not in the table (ACC: absent & exists is dropped)
        if (emp0) { word[u] = (u64)b0[u] * 16 + (__ffs(emp0) - 1); act[u] = 2; }   // first empty slot in probe order
        else if (emp1) { word[u] = (u64)b1[u] * 16 + (__ffs(emp1) - 1); act[u] = 2; flag_b0[u] = !ovf0; }
        else if (fl.spec) {
          // both home buckets full on a table that no longer walks: replace the minimum-score entry of the 30 slots
          if (hf) need_v[u] = true;                   // (its score lines are not here yet: below)
          else choose_victim(u, ovf0);
          ovf0_u[u] = ovf0;
        } else why[u] = 2;   // a table that still walks (not at capacity / unbounded): placed further along by the general path
      }
    }
  }
  if (hf && fl.spec) {
    bool any_v = false;
#pragma unroll
    for (int u = 0; u < U; ++u) any_v = any_v || need_v[u];
    if (__ballot(any_v)) {   // (wave-uniform) one more round trip, for the waves that hold a key in need of a victim
#pragma unroll
      for (int u = 0; u < U; ++u) {
        sc[u][0] = (i64)score_line(v, b0[u])[sub];
        sc[u][1] = (i64)score_line(v, b1[u])[sub];
      }
      keep_live_u2<U>(sc, 0);
      keep_live_u2<U>(sc, 1);
#pragma unroll
      for (int u = 0; u < U; ++u) if (need_v[u]) choose_victim(u, ovf0_u[u]);
    }
  }
  // ---- now the claims -------------------------------------------------------------------------------------------
  const unsigned lostreg = reserved ? 2u : ((c0 == gen || c1 == gen) ? 1u : 0u);   // (group 0's lanes)
  // HF (the overlapped step): a HIT needs no claim.  It writes its own row and score word and nothing else of the bucket; the only
  // writer that could take its slot away is an eviction, and an eviction never takes a key of this batch (the victim check above
  // looks the victim up in the batch's own plan too).  Only the keys that CHANGE a bucket — a new key into a free slot or over a
  // victim — claim their two home buckets, now, behind the decision: 96 % of a Zipf batch's keys issue no atomic at all, two keys of
  // a batch sharing a home bucket no longer collide unless both are new, and the item list is empty in nearly every step.
  bool lost_hf[U];
#pragma unroll
  for (int u = 0; u < U; ++u) lost_hf[u] = false;
  if (hf) {
    bool any = false;
#pragma unroll
    for (int u = 0; u < U; ++u) { lost_hf[u] = false; any = any || act[u] >= 2; }
    if (__ballot(any)) {   // (wave-uniform: one more round trip for the waves that hold a new key)
      unsigned cx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        cx[u] = 0;
        if (act[u] >= 2 && sub < 2) cx[u] = atomicExch(a.tags + (sub == 0 ? b0[u] : b1[u]), gen) == gen ? 1u : 0u;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) lost_hf[u] = ((__ballot(cx[u] != 0) >> gshift) & 0xffffu) != 0;
    }
  }
  unsigned last[U], hint[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int j = u * 4 + grp;
    last[u] = (unsigned)__shfl((int)lastreg, j);
    const bool real = __shfl((int)valid, j) != 0;
    const int lost = hf ? ((real && !on[u]) ? 2 : (lost_hf[u] ? 1 : 0)) : __shfl((int)lostreg, j);   // (without HF: lane j of group 0 made the claims)
    hint[u] = 0;
    if (lost) {
      if (lost == 1 && act[u] == 1 && (word[u] >> 32) == 0) hint[u] = 1;   // found, not written: the remainder pass locks this very slot
      act[u] = 0; why[u] = lost;
    }
    if (!hf && bxc[u] != ~0u && act[u]) {   // (rare) found beyond its home buckets: that bucket's claim
      unsigned cx = 0;
      if (sub == 0) cx = atomicExch(a.tags + bxc[u], gen) == gen ? 1u : 0u;
      if (__shfl((int)cx, gshift)) { act[u] = 0; why[u] = 1; }
    }
    if (!real) { act[u] = 0; why[u] = 0; }
    if (ACC) {
      if (act[u] == 1 && !ex[u]) act[u] = 0;                                   // present & !exists: dropped
      if (why[u] == 1 && hint[u] && !ex[u]) { why[u] = 0; hint[u] = 0; }        // (the same, seen without the claim)
      if (why[u] == 1 && !hint[u] && ex[u] && bxc[u] == ~0u) { }               // lost its claim, not seen: the remainder looks again
      if (ex[u]) hint[u] |= 2u;                                                // the item carries the flag
    }
    if (act[u] && flag_b0[u] && sub == 15) atomicOr((u64*)(key_line(v, b0[u]) + 15), META_OVF0);   // finds go on to b1
    if (sub == 0 && why[u]) { if (CF) __hip_atomic_store(a.dflag + gk[u], (uint8_t)4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else a.dflag[gk[u]] = 4; }
    fresh += (act[u] == 2 && sub == 0);
    if (not_hits) *not_hits += (real && act[u] != 1 && sub == 0);   // (a new key, an eviction, a key handed to the remainder)
  }
  {   // left-over keys of the wave -> the list: one atomic add for all of them
    u64 sm[U];
    unsigned nslow = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) { sm[u] = __ballot(why[u] != 0 && sub == 0); nslow += (unsigned)__popcll(sm[u]); }
    if (nslow) {
      unsigned at = 0;
      if (lane == 0) at = atomicAdd(slow_ctr, nslow);
      at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (why[u] != 0 && sub < 2) {
          const unsigned pos = at + (unsigned)__popcll(sm[u] & ((1ULL << gshift) - 1));
          if (pos < a.item_cap) {
            uint4 w;
            if (sub == 0) w = make_uint4((unsigned)(u64)key[u], (unsigned)((u64)key[u] >> 32), last[u], gk[u]);
            else w = make_uint4((unsigned)in_s[u], (unsigned)(in_s[u] >> 32), hint[u], (unsigned)word[u]);
            if (CF) store_wt16(reinterpret_cast<unsigned char*>(a.items + pos) + sub * 16, w);   // (read by the tail role of the same launch)
            else *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(a.items + pos) + sub * 16) = w;
          }
        }
        at += (unsigned)__popcll(sm[u]);
      }
    }
  }
  // value rows of the 4 keys: loads together (always from a valid address), stores for the keys that write
  typedef typename Granule<G>::T T;
  unsigned char* dst[U];
#pragma unroll
  for (int u = 0; u < U; ++u) dst[u] = row_at(v, word[u] >> 4, (unsigned)word[u] & 15u);
  for (unsigned off = sub * G; off < v.field_bytes; off += 16 * G) {
    T tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) tmp[u] = *reinterpret_cast<const T*>(a.vals + (u64)last[u] * (u64)v.field_bytes + off);
    keep_live_u<U>(tmp);
    if (ACC && G == 16) {   // accumulate: the rows themselves travel with the deltas (a key that only inserts adds nothing)
      T cur[U];
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = *reinterpret_cast<const T*>(dst[u] + off);
      keep_live_u<U>(cur);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (act[u] == 1) *reinterpret_cast<uint4*>(&tmp[u]) = add16_dt(*reinterpret_cast<uint4*>(&cur[u]), *reinterpret_cast<uint4*>(&tmp[u]), a.acc_dt);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!act[u]) continue;
      // write-through: the rows leave L2 during the kernel instead of at the boundary to the next one
      if (G == 16) store_wt16(dst[u] + off, *reinterpret_cast<uint4*>(&tmp[u]));
      else __hip_atomic_store(reinterpret_cast<T*>(dst[u] + off), tmp[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (!act[u]) continue;
    if (act[u] >= 2) {
      if (!SIMPLE && v.n_fields > 1) {
        for (unsigned f = 1; f < v.n_fields; ++f) {   // slot fields of a brand-new row start at aux_init
          const unsigned pat = a.ai.pattern[(f - 1) & 3];
          unsigned char* q = dst[u] + f * v.field_bytes;
          if ((v.field_bytes & 3) == 0)
            for (unsigned off = sub * 4; off < v.field_bytes; off += 64)
              __hip_atomic_store(reinterpret_cast<unsigned*>(q + off), pat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            for (unsigned off = sub; off < v.field_bytes; off += 16)
              __hip_atomic_store(q + off, (unsigned char)(pat >> (8 * (off % a.ai.elem_bytes))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (sub == 0) { if (CF) store_wt8(key_word(v, word[u]), (u64)key[u]); else *key_word(v, word[u]) = key[u]; }   // owned bucket: a plain store (CF: write-through, the
                                                                                                                   // left-over keys follow in the same launch)
    }
    if (!fl.with_scores) continue;
    if (fl.lru) { if (sub == 0) { if (CF) store_wt8(score_word(v, word[u]), now); else *score_word(v, word[u]) = now; } }
    else if (act[u] == 3 && a.sp.strategy == TFRA_EVICT_LFU) { if (sub == 0) store_wt8(score_word(v, word[u]), in_s[u]); }   // the slot starts a new life
    else update_score<true>(v, (i64)((word[u] >> 4) * SLOTS + (word[u] & 15)), act[u] >= 2, a.sp.strategy, in_s[u], a.sp.epoch, sub);
  }
}

// HF (SRC_SET, round 6 — what the overlapped step's write-back role has done since round 5): a HIT claims nothing and reads no score
// line; a key that changes a bucket claims behind its decision, and a victim that is a key of this very batch (a probe of the batch's own
// SET plan) sends the new key to the remainder.  On a Zipf batch over resident ids (96 % hits) a key then touches its two key lines, its
// value row and its row instead of four lines, two claim words and the rows; a batch of mostly NEW keys pays one more dependent trip per
// wave (the score lines, then the claims) — the host picks the form from a sample of the previous write-back (launch_own).
// SAMPLE: the launch leaves the sample described below (compiled out where nobody reads it: a caller's keys on a table that evicts)
template <int G, bool SIMPLE, int SRC, int U = 4, bool ACC = false, bool HF = false, bool SAMPLE = false>
__global__ __launch_bounds__(256) void upsert_own_kernel(const OwnArgs a, OwnCtrs* ctr, unsigned own_gen, unsigned* progress,
                                                         unsigned progress_val) {
  const int lane = threadIdx.x & 63;
  const unsigned total = SRC != SRC_DIRECT ? a.ks.d_counts[0] + a.ks.d_counts[1] : direct_count(a);
  const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int fresh = 0, not_hits = 0, looked = 0;
  // every 128th wave tells how many of its keys were not plain hits — ONE 64-bit add per such wave (every 16th wave with two adds: ~180
  // same-line atomics piling up at the end of a one-round kernel, 9.9 -> 12.4 us for the DIRECT pass of 22 K keys under rocprofv3)
  const bool sampled = SAMPLE && (wave & 127u) == 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (progress) {   // see hot_sums_kernel; [1]: the keys of this write-back — the host sizes the next one's grid from it
      __hip_atomic_store(progress, progress_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(progress + 1, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (SRC == SRC_PLAN && a.ks.d_counts[5]) atomicAdd(a.v.err_count, a.ks.d_counts[5]);
  }
  const OwnFlags fl = own_setup<SIMPLE>(a);
  for (unsigned wbase = wave * (4 * U); wbase < total; wbase += nwaves * (4 * U)) {
    const unsigned i = wbase + (unsigned)(lane & 15);
    const bool valid = (lane & 15) < 4 * U && i < total;
    own_batch16<G, SIMPLE, SRC, U, false, ACC, HF>(a, fl, min(i, total - 1), valid, own_gen, &ctr->n_a, lane, fresh, nullptr, nullptr, 0, 0,
                                                   (HF && a.own_set.ent) ? &a.own_set : nullptr, (SAMPLE && sampled) ? &not_hits : nullptr);
    if (SAMPLE) looked += (valid && lane < 16);
  }
  for (int off = 32; off > 0; off >>= 1) fresh += __shfl_xor(fresh, off);
  if (lane == 0 && fresh) size_add(a.v, wave, fresh);
  if (SAMPLE && sampled) {
    for (int off = 32; off > 0; off >>= 1) { not_hits += __shfl_xor(not_hits, off); looked += __shfl_xor(looked, off); }
    if (lane == 0 && looked)   // spare[1] (low word: keys looked at) | spare[2] (high word: not plain hits), 8-byte aligned
      atomicAdd(reinterpret_cast<unsigned long long*>(&ctr->spare[1]), ((unsigned long long)(unsigned)not_hits << 32) | (unsigned long long)(unsigned)looked);
  }
}

// ---------------------------------------------------------------------------------------------
// SET plan: the id-only half of an ASSIGN write-back (tfra_sparse_plan_build with dim 0) in ONE kernel.
// An assign needs, per distinct id, the position of its LAST occurrence (and how often it occurred: LFU scores) — not the
// CSR of all positions the gradient sums need.  The CSR plan's three kernels take 29 us for 131 072 ids and six of the
// step's nine kernel launches; the step driver that builds it one batch ahead was bound by the host's launch rate
// (52 us per step of host time for 39 us of kernels on the main stream).
//   phase A  one block per 1024 ids: equal ids meet in an LDS hash table (compare-and-swap on the key word, atomic max on
//            position + 1, atomic add on the count);
//   phase B  every distinct id of the block goes into a global open-addressing table of >= 2 n slots (compare-and-swap on
//            the key, atomic max / add on (position + 1, count)): its slot is the same for every block.  The block whose
//            swap installed the key appends (key, slot) to the dense list of distinct keys — one counter add per block;
//   phase C  the plan keeps TWO such tables and alternates: while build k fills one, it empties the slots build k-1 used in
//            the other (its dense list says which) — no memset, no extra launch.
// NO block finishes the build for the others (round 3: the kernel used to end with a ticket — wait for the block's stores,
// draw, the last block reads the count and publishes it to device and pinned memory, resets the counters: four more
// dependent round trips, 6 of the kernel's 17.8 us).  The count of distinct keys IS the append counter, read by the
// consumer after the kernel; each table has two counter words and alternates between them from use to use — block 0 of a
// build zeroes the word of the table's NEXT use (its last reader, the other table's build right after the use before,
// is over by stream order).  The host sizes the consumer's grid from the count the previous write-back saw
// (upsert_own_kernel publishes it), grid-stride covers a batch that has more.
// The two sentinel key values have slots of their own behind the table (no hashing: EMPTY_KEY is the free-slot marker).
constexpr int SP_NT = 1024;
constexpr unsigned SP_LDS = 2048;
constexpr unsigned SET_PAD = 4;   // entries behind the two sentinel slots, never used: a probe reads 4 consecutive entries
struct SetTab {   // one of the plan's two tables
  SetEnt* ent;      // [m2 + 2 + SET_PAD]  key: EMPTY_KEY = free (the two sentinel slots [m2], [m2 + 1]: EMPTY_KEY = free, else taken)
  i64* ukeys;       // [n] dense list: the distinct keys, in no particular order
  unsigned* uslot;  // [n] their slots
  unsigned* count;  // number of distinct keys of the table's current use (zero before its build)
};

// COUNTS: also count the occurrences of every id (LFU scores without caller scores; tfra_sparse_plan_read)
// INDEX: the occurrence-count word of a key's table entry receives the key's position in the dense list instead (tfra_unique_unordered)
// FUSED (tfra_unique_unordered in ONE launch, INDEX only, grids of <= 128 blocks — all co-resident on half the chip): the kernel also
// writes the inverse index idx_out[i] = position of ids[i] in the dense list, the list itself into unique_out and — the last block to
// add its share — its length into num_out.  A key's dense index exists once the block that INSTALLED the key has drawn its base
// from the append counter; the other blocks holding the key poll the entry's index word (index + 1, 0 = not yet) — one poller per
// block and distinct id (the block's ids share the answer through LDS), a wait of one atomic's round trip.
// IPT ids per thread (1; 2 in find_unique_kernel: half the blocks — half the wave slots — for the same ids; the LDS table grows with it)
template <bool COUNTS, bool INDEX, bool FUSED, int IPT = 1>
__device__ __forceinline__ void setplan_block(const unsigned bid, const unsigned nblk, size_t n, const i64* __restrict__ ids, unsigned m2,
                                              const SetTab& cur, const SetTab& old, unsigned* next_use_count, i64* __restrict__ unique_out,
                                              int* __restrict__ idx_out, i64* __restrict__ num_out, unsigned* err = nullptr) {
  static_assert(!FUSED || (INDEX && !COUNTS), "FUSED: the unique-with-index build");
  constexpr unsigned LDSN = SP_LDS * IPT;   // slots of the block's LDS table: two per id
  constexpr int NR = 2 * IPT;               // ... = NR per thread
  __shared__ i64 s_key[LDSN];
  __shared__ unsigned s_pos[LDSN + 2], s_cnt[LDSN + 2];
  __shared__ unsigned s_n, s_base;
  const unsigned tid = threadIdx.x;
  const unsigned n_old = *old.count;
  if (bid == 0 && tid == 0) *next_use_count = 0;
  for (unsigned i = tid; i < LDSN + 2; i += SP_NT) { if (i < LDSN) s_key[i] = EMPTY_KEY; s_pos[i] = 0; if (COUNTS || FUSED) s_cnt[i] = 0; }
  if (tid == 0) s_n = 0;
  __syncthreads();
  // ---- A: equal ids of the block meet in LDS ---------------------------------------------------------------
  const size_t gid = (size_t)bid * SP_NT + tid;   // (phase C's start)
  size_t gids[IPT];
  unsigned lds_slot[IPT];   // FUSED: where each of this thread's ids sits in the block's LDS table
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    gids[q] = ((size_t)bid * IPT + q) * SP_NT + tid;
    lds_slot[q] = 0;
    if (gids[q] < n) {
      const i64 id = ids[gids[q]];
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
      atomicMax(&s_pos[slot], (unsigned)gids[q] + 1u);
      if (COUNTS) atomicAdd(&s_cnt[slot], 1u);
      lds_slot[q] = slot;
    }
  }
  __syncthreads();
  // ---- B: the block's distinct ids into the global table: the first probes of both of a thread's keys travel together ----
  i64 mykey[NR];
  unsigned myslot[NR], myidx[NR], p1[NR], cn[NR];
  bool have[NR], mine[NR];
  i64 was[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const unsigned s = tid + (unsigned)r * SP_NT;   // 2048 hash slots per id of a thread; the two sentinel slots ride with threads 0 and 1 below
    mine[r] = false;
    mykey[r] = s_key[s];
    p1[r] = s_pos[s]; cn[r] = COUNTS ? s_cnt[s] : 0u;
    have[r] = p1[r] != 0;
    myslot[r] = (unsigned)(fmix64((u64)mykey[r]) >> 20) & (m2 - 1);
    was[r] = 0;
    if (have[r]) was[r] = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&cur.ent[myslot[r]].key), (unsigned long long)EMPTY_KEY, (unsigned long long)mykey[r]);
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if (have[r]) {
      for (;;) {
        if (was[r] == EMPTY_KEY) { mine[r] = true; break; }
        if (was[r] == mykey[r]) break;
        myslot[r] = set_at(myslot[r], 1u, set_wmask(m2));
        was[r] = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&cur.ent[myslot[r]].key), (unsigned long long)EMPTY_KEY, (unsigned long long)mykey[r]);
      }
      // (INDEX — tf.unique: nobody reads a last position, and the entry's other word is what the other blocks POLL for the key's index:
      // an atomic on the line they are reading is the slow kind, NOTEBOOK round 2)
      if (!INDEX) atomicMax(&cur.ent[myslot[r]].pos1, p1[r]);
      if (COUNTS) atomicAdd(&cur.ent[myslot[r]].cnt, cn[r]);
    }
    myidx[r] = mine[r] ? atomicAdd(&s_n, 1u) : 0u;
  }
  if (tid < 2 && s_pos[LDSN + tid] != 0) {   // a sentinel key value occurred in this block
    const unsigned sl = m2 + tid;
    const i64 w = (i64)atomicCAS(reinterpret_cast<unsigned long long*>(&cur.ent[sl].key), (unsigned long long)EMPTY_KEY, 1ULL);
    if (!INDEX) atomicMax(&cur.ent[sl].pos1, s_pos[LDSN + tid]);
    if (COUNTS) atomicAdd(&cur.ent[sl].cnt, s_cnt[LDSN + tid]);
    if (w == EMPTY_KEY) {
      const unsigned at = atomicAdd(cur.count, 1u);   // (rare: its own add)
      cur.ukeys[at] = EMPTY_KEY + (i64)tid;
      cur.uslot[at] = sl;
      if (FUSED) { unique_out[at] = EMPTY_KEY + (i64)tid; __hip_atomic_store(&cur.ent[sl].cnt, at + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else if (INDEX) cur.ent[sl].cnt = at;
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (FUSED) {
      // the block's share and its ARRIVAL in one 64-bit add — the word in front of the count (always 0 otherwise: CsrKeys::d_counts[0])
      // counts the blocks: the last one to arrive sees the others' shares in the value returned and knows the list's length there
      // and then (a ticket of its own at the end of the kernel was one more round trip); it leaves the word at 0 again
      const unsigned long long r = atomicAdd(reinterpret_cast<unsigned long long*>(cur.count - 1), ((unsigned long long)s_n << 32) | 1ULL);
      s_base = (unsigned)(r >> 32);
      if ((unsigned)r == nblk - 1u) {
        *num_out = (i64)(s_base + s_n);
        __hip_atomic_store(cur.count - 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      s_base = s_n ? atomicAdd(cur.count, s_n) : 0u;
    }
  }
  // ---- C: empty the slots the previous build used in the OTHER table (while the counter add travels) -------------------
  for (size_t i = gid; i < n_old; i += (size_t)nblk * SP_NT) {
    const unsigned sl = old.uslot[i];
    *reinterpret_cast<uint4*>(old.ent + sl) = make_uint4(0u, 0x80000000u, 0u, 0u);   // {EMPTY_KEY, 0, 0}
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if (!mine[r]) continue;
    cur.ukeys[s_base + myidx[r]] = mykey[r];
    cur.uslot[s_base + myidx[r]] = myslot[r];
    if (FUSED) { unique_out[s_base + myidx[r]] = mykey[r]; __hip_atomic_store(&cur.ent[myslot[r]].cnt, s_base + myidx[r] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else if (INDEX) cur.ent[myslot[r]].cnt = s_base + myidx[r];
  }
  if (FUSED) {
    // the dense index of every distinct id of the block -> LDS (s_cnt is free here): drawn above for the keys this block installed,
    // polled from the entry for the others (their installer is a resident block: at most 128 de-duplicating blocks — 256 with
    // TFRA_FU_IPT=1 above 131072 ids, still co-resident: 1024 threads of 57 registers, four blocks per CU).  The poll is bounded; a
    // timeout leaves idx_out = -1 for the block's positions of that id AND counts an error (`err`: the table's counter under
    // tfra_table_find_unique, the workspace's under tfra_unique_unordered), so that nothing indexes with it unnoticed.
    bool timed_out = false;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (!have[r]) continue;
      unsigned v = mine[r] ? s_base + myidx[r] + 1u : 0u;
      for (unsigned it = 0; !v && it < (1u << 24); ++it) {
        v = __hip_atomic_load(&cur.ent[myslot[r]].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!v) __builtin_amdgcn_s_sleep(2);
      }
      timed_out |= v == 0u;
      s_cnt[tid + (unsigned)r * SP_NT] = v;
    }
    if (tid < 2 && s_pos[LDSN + tid] != 0) {   // a sentinel key value occurred in this block
      unsigned v = 0;
      for (unsigned it = 0; !v && it < (1u << 24); ++it) {
        v = __hip_atomic_load(&cur.ent[m2 + tid].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!v) __builtin_amdgcn_s_sleep(2);
      }
      timed_out |= v == 0u;
      s_cnt[LDSN + tid] = v;
    }
    if (timed_out && err) atomicAdd(err, 1u);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IPT; ++q)
      if (gids[q] < n) idx_out[gids[q]] = (int)s_cnt[lds_slot[q]] - 1;   // (-1 only after a poll that timed out: never seen; counted in *err)
  }
}

template <bool COUNTS, bool INDEX = false, bool FUSED = false>
__global__ __launch_bounds__(SP_NT) void setplan_kernel(size_t n, const i64* __restrict__ ids, unsigned m2, SetTab cur, SetTab old,
                                                        unsigned* next_use_count, i64* __restrict__ unique_out = nullptr,
                                                        int* __restrict__ idx_out = nullptr, i64* __restrict__ num_out = nullptr,
                                                        unsigned* err = nullptr) {
  setplan_block<COUNTS, INDEX, FUSED>(blockIdx.x, gridDim.x, n, ids, m2, cur, old, next_use_count, unique_out, idx_out, num_out, err);
}

// TFRA>HkvHashTableEmbeddingLookup in ONE launch (tfra_table_find_unique): the first `ublocks` blocks (<= 128: co-resident, see FUSED)
// de-duplicate the ids — distinct ids, inverse index, count —, the blocks behind them look the SAME ids up in the table.  The
// de-duplication waits (atomics' round trips: 85 % of its wave cycles), the lookup moves bytes: side by side they take the time of
// the longer one instead of the sum plus a launch gap.  16 waves per block, the lookup's waves as in find_kernel.
template <int G, bool PF1, int FU_IPT>
__global__ __launch_bounds__(SP_NT) void find_unique_kernel(unsigned ublocks, size_t n, const i64* __restrict__ ids, unsigned m2, SetTab cur, SetTab old,
                                                            unsigned* next_use_count, i64* __restrict__ unique_out, int* __restrict__ idx_out,
                                                            i64* __restrict__ num_out, TableView v, unsigned char* __restrict__ rows_out,
                                                            uint8_t* __restrict__ exists, const unsigned char* __restrict__ defaults, int full) {
  if (blockIdx.x < ublocks) {
    setplan_block<false, true, true, FU_IPT>(blockIdx.x, ublocks, n, ids, m2, cur, old, next_use_count, unique_out, idx_out, num_out, v.err_count);
    return;
  }
  find_wave<G, 4, G == 16, PF1>(v, n, ids, rows_out, exists, defaults, full, 0u, (blockIdx.x - ublocks) * (SP_NT / 64) + (threadIdx.x >> 6));
}

// tfra_unique_unordered, second launch: idx[i] = position of ids[i] in the plan's dense list (a probe of the plan's table);
// the list itself and its length are copied out on the way.
__global__ __launch_bounds__(256) void unique_idx_kernel(size_t n, const i64* __restrict__ ids, SetProbe pr, const i64* __restrict__ ukeys,
                                                         const unsigned* __restrict__ count, i64* __restrict__ unique_out, int* __restrict__ idx_out,
                                                         i64* __restrict__ num_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned U = *count;
  if (i == 0) *num_out = (i64)U;
  if (i < U) unique_out[i] = ukeys[i];
  if (i >= n) return;
  const i64 id = ids[i];
  unsigned slot = set_home(pr, id, fmix64((u64)id));
  const unsigned wm = set_wmask(pr.m2);
  int found = -1;
  if (is_reserved_key(id)) found = (int)pr.ent[slot].cnt;
  else {
    for (unsigned g = 0; g <= wm; ++g) {
      const SetEnt* e = pr.ent + set_at(slot, g, wm);
      const i64 k = e->key;
      if (k == id) { found = (int)e->cnt; break; }
      if (k == EMPTY_KEY) break;
    }
  }
  idx_out[i] = found;
}

#define TFRA_STEP_DEVICE_PART
#include "tfra_step_impl.h"
#undef TFRA_STEP_DEVICE_PART

__global__ void fill_setent_kernel(SetEnt* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0x80000000u, 0u, 0u);   // {EMPTY_KEY, 0, 0}
}
__global__ void fill_i64_kernel(i64* p, size_t n, i64 v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
struct tfra_sparse_plan {
  int device = 0;
  void* buf = nullptr;
  size_t bytes = 0;
  // layout of the last build
  size_t n = 0, npad = 0, ntiles = 0;
  unsigned P = 0, cm = 0;
  int dim = 0;
  unsigned* cursors = nullptr;
  CsrDesc ds{};
  CsrOut out{};
  unsigned* tile_entries = nullptr;
  unsigned short* run_start = nullptr;
  unsigned* tile_len = nullptr;
  uint4* drec = nullptr;
  unsigned* keymap = nullptr;
  i64* dkeys = nullptr;
  unsigned* binmap = nullptr;
  unsigned* d_counts = nullptr;
  uint8_t* dflag = nullptr;
  size_t dflag_len = 0;
  OwnItem* slow_items = nullptr;   // [SLOW_CAP] left-over keys of the ownership pass of a write-back (self-contained items)
  unsigned* any_deferred = nullptr;   // = use_gen of the last write-back that deferred a key to its eviction phase
  mutable unsigned use_gen = 0;
  mutable unsigned ups_uses[2] = {0, 0};   // upsert_planned uses of the CSR buffer / of the SET buffer: the parity selects the left-over
                                           // counter set of THAT buffer (each buffer has its own two sets and its own flag bytes: a plan
                                           // object rebuilt with dim > 0, then dim 0, then dim > 0 again must find its CSR buffer's sets
                                           // where its last CSR use left them)
  float* partial = nullptr;
  int* prow_dest = nullptr;        // [partial rows] scratch of tfra_plan_positions_to
  bool armed = false;              // cursors/counters are zero (re-armed by the last kernel of the previous build)
  unsigned* host_counts = nullptr; // pinned: [0] generation of the last COMPLETED build, [1..6] its counts
  unsigned gen = 0;                // generation of the last enqueued build
  bool ev_recorded = false;        // the last build ran on a side stream (tfra_table_step_prefetch)
  unsigned last_used_step = 0;     // last step whose write-back read this plan
  hipEvent_t built_ev = nullptr;   // recorded behind a build on a side stream (tfra_table_step_prefetch): complete = the plan is in memory
  // SET plan (dim 0): its own buffer; two tables alternate (setplan_kernel)
  int kind = 0;                    // 0 CSR, 1 SET
  void* setbuf = nullptr;
  size_t set_cap = 0;              // ids the set buffer was sized for
  unsigned set_m2 = 0;
  SetTab set_tab[2]{};
  unsigned set_parity = 0;         // table of the last build
  unsigned* set_counts = nullptr;  // the d_counts block of the set buffer: [8] any_deferred [12..19] OwnCtrs x2; [64 + 8 (2 p + use & 1) ..]
                                   // the key counts of table p's uses ({0, distinct keys, 0, 0, 0, 0}: what CsrKeys::d_counts shows)
  unsigned set_use[2] = {0, 0};    // uses of each table so far
  bool built_counts = true;        // the last build counted occurrences
  bool skip_counts_once = false;   // the NEXT build need not count occurrences (set by the table's own drivers for tables whose
                                   // scores do not read them; a build through the public entry point always counts)
  uint8_t* set_dflag = nullptr;
  OwnItem* set_items = nullptr;
  // the overlapped step (tfra_step_impl.h) builds a SET plan in two launches without atomics: launch 1 scatters every tile's
  // distinct (id, last position) pairs into per-window segments, launch 2 builds each window of the table from its segments
  void* segbuf = nullptr;
  size_t seg_cap_ids = 0;          // ids the scatter buffers were sized for
  SetEnt* seg_pairs = nullptr;     // [windows][tiles][SEG_CAP]
  unsigned* seg_cnt = nullptr;     // [windows][tiles]
  SetEnt* ovf_pairs = nullptr;     // pairs that did not fit their segment (an adversarial batch): appended with an atomic
  unsigned* ovf_cnt = nullptr;
  unsigned* ucnt = nullptr;        // [windows <= 256] distinct keys per window of the last list-less build (the step launch's BUILD role)
  unsigned seg_tiles = 0;          // tiles of the last scatter
  unsigned scat_use = 0;           // scatters into this object so far (its two overflow counters alternate)
  const int64_t* scat_ids = nullptr;   // the batch whose pairs the segments hold (nullptr: none)
  size_t scat_n = 0;
  unsigned char tab_state[2] = {0, 0}; // TAB_EMPTY / TAB_LISTED / TAB_LISTLESS: what each table holds (setplan_prepare)
};

extern "C" int tfra_sparse_plan_create(int device, tfra_sparse_plan_t** out) {
  if (!out) return set_error(TFRA_ERR_INVALID, "sparse_plan_create: null out");
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_create: no device");
  tfra_sparse_plan* pl = new tfra_sparse_plan();
  pl->device = device;
  *out = pl;
  return TFRA_OK;
}

extern "C" int tfra_sparse_plan_destroy(tfra_sparse_plan_t* pl) {
  if (!pl) return TFRA_OK;
  if (pl->buf) { (void)hipSetDevice(pl->device); (void)hipDeviceSynchronize(); (void)hipFree(pl->buf); }
  if (pl->setbuf) { (void)hipSetDevice(pl->device); (void)hipDeviceSynchronize(); (void)hipFree(pl->setbuf); }
  if (pl->segbuf) { (void)hipSetDevice(pl->device); (void)hipDeviceSynchronize(); (void)hipFree(pl->segbuf); }
  if (pl->host_counts) (void)hipHostFree(pl->host_counts);
  if (pl->built_ev) (void)hipEventDestroy(pl->built_ev);
  delete pl;
  return TFRA_OK;
}

static size_t plan_smem_bytes(unsigned cm) { return (size_t)cm * 36 + (size_t)TABW * 4; }

// The SET plan of a batch (dim 0): see setplan_kernel.  setplan_prepare = everything but the launch (buffers, which of the two
// tables, its counter words): the overlapped step builds the plan inside its own kernel (the BUILD / SCATTER roles of tfra_step_impl.h).
struct SetPlanLaunch { SetTab cur, old; unsigned* next_use_count; unsigned m2; unsigned blocks; };
static int setplan_ensure(tfra_sparse_plan* pl, size_t n, hipStream_t s) {   // the SET buffer, sized for n ids
  if (n > MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "sparse_plan_build: at most 2^18 ids per plan");
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  if (pl->set_cap < n) {
    if (pl->setbuf) {
      if (hipDeviceSynchronize() != hipSuccess || hipFree(pl->setbuf) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: free");
      pl->setbuf = nullptr; pl->set_cap = 0;
    }
    const size_t cap = std::max<size_t>(n, 4096);
    unsigned m2 = 4096;
    while ((size_t)m2 < 2 * cap) m2 <<= 1;
    const size_t tab = al(((size_t)m2 + 2 + SET_PAD) * sizeof(SetEnt)) + al(cap * 8) + al(cap * 4);   // entries, ukeys, uslot
    const size_t bytes = 512 + al((size_t)m2 + 8) + al((size_t)SLOW_CAP * sizeof(OwnItem)) + 2 * tab;   // (flag bytes: per key of a list, or per SLOT for the overlapped step)
    hipError_t e = hipMalloc(&pl->setbuf, bytes);
    if (e != hipSuccess) { pl->setbuf = nullptr; return set_error(e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP, "sparse_plan_build: hipMalloc"); }
    if (hipMemsetAsync(pl->setbuf, 0, bytes, s) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: memset");
    unsigned char* w = (unsigned char*)pl->setbuf;
    pl->set_counts = (unsigned*)w; w += 512;          // [0..5] counts [6] ticket [8] any_deferred [12..19] OwnCtrs x2 [20,21] list counts [32] i64 count
    pl->set_dflag = (uint8_t*)w; w += al((size_t)m2 + 8);
    pl->set_items = (OwnItem*)w; w += al((size_t)SLOW_CAP * sizeof(OwnItem));
    for (int p = 0; p < 2; ++p) {
      SetTab& tb = pl->set_tab[p];
      tb.ent = (SetEnt*)w; w += al(((size_t)m2 + 2 + SET_PAD) * sizeof(SetEnt));
      tb.ukeys = (i64*)w; w += al(cap * 8);
      tb.uslot = (unsigned*)w; w += al(cap * 4);
      tb.count = nullptr;   // set per build: the table's two counter words alternate
      fill_setent_kernel<<<256, 256, 0, s>>>(tb.ent, (size_t)m2 + 2 + SET_PAD);
    }
    pl->set_cap = cap; pl->set_m2 = m2; pl->set_parity = 1; pl->set_use[0] = pl->set_use[1] = 0;
    pl->tab_state[0] = pl->tab_state[1] = 0;
  }
  return TFRA_OK;
}
// What each of the object's two tables holds (tab_state): the build rules differ in what they leave behind —
//   a REGULAR build (setplan_kernel) needs an EMPTY target, leaves it LISTED (keys + the dense list of the slots they took) and
//     empties the OTHER table through that table's list;
//   a LIST-LESS build (the step launch's BUILD role) rewrites every slot of its target, leaves it LISTLESS and touches nothing else.
// A plan object may see them in any order (a step driver whose look-ahead is sometimes missing: regular, list-less, regular on one
// object left the third build's target holding the first batch's keys).  setplan_prepare makes both tables what the regular
// build expects, whatever came before: a target that is not known empty is filled, an other table without a list is filled
// instead of walked, and a filled table's counter words are zeroed (a list-less build zeroes none).
enum : unsigned char { TAB_EMPTY = 0, TAB_LISTED = 1, TAB_LISTLESS = 2 };
static int setplan_fill_table(tfra_sparse_plan* pl, unsigned q, hipStream_t s) {
  fill_setent_kernel<<<256, 256, 0, s>>>(pl->set_tab[q].ent, (size_t)pl->set_m2 + 2 + SET_PAD);
  if (hipMemsetAsync(pl->set_counts + 64 + 16 * q, 0, 16 * sizeof(unsigned), s) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: memset");
  pl->tab_state[q] = TAB_EMPTY;
  return TFRA_OK;
}
static int setplan_prepare(tfra_sparse_plan* pl, size_t n, hipStream_t s, bool counts, SetPlanLaunch* L) {
  int rc = setplan_ensure(pl, n, s);
  if (rc) return rc;
  const unsigned p = pl->set_parity ^ 1u;
  pl->gen += 1;
  const unsigned blocks = (unsigned)((n + SP_NT - 1) / SP_NT);
  auto count_word = [&](unsigned tab, unsigned use) { return pl->set_counts + 64 + 8 * (2 * tab + (use & 1u)) + 1; };
  if (pl->tab_state[p] != TAB_EMPTY && (rc = setplan_fill_table(pl, p, s)) != TFRA_OK) return rc;
  if (pl->tab_state[p ^ 1u] == TAB_LISTLESS && (rc = setplan_fill_table(pl, p ^ 1u, s)) != TFRA_OK) return rc;
  const unsigned use = ++pl->set_use[p];
  SetTab cur = pl->set_tab[p], old = pl->set_tab[p ^ 1u];
  cur.count = count_word(p, use);
  // the other table is walked through its list only when it has one; otherwise (never used, just filled) the list is empty: a word that is always zero
  old.count = pl->tab_state[p ^ 1u] == TAB_LISTED ? count_word(p ^ 1u, pl->set_use[p ^ 1u]) : pl->set_counts + 120;
  pl->set_tab[p].count = cur.count;
  pl->tab_state[p] = TAB_LISTED; pl->tab_state[p ^ 1u] = TAB_EMPTY;
  pl->scat_ids = nullptr; pl->scat_n = 0;   // (pairs a step launch scattered for this object belong to a batch this build replaces)
  L->cur = cur; L->old = old; L->next_use_count = count_word(p, use + 1); L->m2 = pl->set_m2; L->blocks = blocks;
  pl->built_counts = counts;
  pl->set_parity = p;
  pl->d_counts = pl->set_counts; pl->dflag = pl->set_dflag; pl->slow_items = pl->set_items; pl->any_deferred = pl->set_counts + 8;
  pl->n = n; pl->dim = 0; pl->kind = 1;
  return TFRA_OK;
}
// The same for a build WITHOUT the dense list and without atomics (the overlapped step: scatter launch + build launch): the
// table that takes the build (every slot of it is written by the build launch), the scatter buffers sized for n ids.
// (The table is sized for at least 131 072 ids — 128 windows — whatever the batch: a scatter tile holds up to 1024 DISTINCT ids, spread over
// the windows into segments of SEG_CAP = 32 entries.  A batch of 22 K all-distinct ids — what one rank of a sharded table serves —
// sized by its own length had 32 windows: 32 ids per segment on average, half the segments overflowed, and the launch took 36 us
// instead of 17: scripts/mb_owner_step.py.)
constexpr size_t LISTLESS_MIN_IDS = 131072;
static int setplan_prepare_listless(tfra_sparse_plan* pl, size_t n, hipStream_t s) {
  int rc = setplan_ensure(pl, std::max(n, LISTLESS_MIN_IDS), s);
  if (rc) return rc;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  if (pl->seg_cap_ids < pl->set_cap) {
    if (pl->segbuf) { if (hipDeviceSynchronize() != hipSuccess || hipFree(pl->segbuf) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: free"); pl->segbuf = nullptr; }
    const size_t wins = pl->set_m2 / SET_WIN, tiles = (pl->set_cap + 1023) / 1024;
    const size_t bytes = 256 + 1024 + al(wins * tiles * SEG_CAP * sizeof(SetEnt)) + al(wins * tiles * 4) + al(pl->set_cap * sizeof(SetEnt));
    hipError_t e = hipMalloc(&pl->segbuf, bytes);
    if (e != hipSuccess) { pl->segbuf = nullptr; return set_error(e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP, "sparse_plan_build: hipMalloc"); }
    if (hipMemsetAsync(pl->segbuf, 0, bytes, s) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: memset");
    unsigned char* w = (unsigned char*)pl->segbuf;
    pl->ovf_cnt = (unsigned*)w; w += 256;
    pl->ucnt = (unsigned*)w; w += 1024;
    pl->seg_pairs = (SetEnt*)w; w += al(wins * tiles * SEG_CAP * sizeof(SetEnt));
    pl->seg_cnt = (unsigned*)w; w += al(wins * tiles * 4);
    pl->ovf_pairs = (SetEnt*)w;
    pl->seg_cap_ids = pl->set_cap;
  }
  return TFRA_OK;
}
// ... and the bookkeeping of the build launch: the object's other table takes the build
static SetTab setplan_take_listless(tfra_sparse_plan* pl, size_t n) {
  const unsigned p = pl->set_parity ^ 1u;
  pl->gen += 1;
  const unsigned use = ++pl->set_use[p];
  pl->set_tab[p].count = pl->set_counts + 64 + 8 * (2 * p + (use & 1u)) + 1;   // (unused by a list-less build: kept valid)
  pl->tab_state[p] = TAB_LISTLESS;
  pl->built_counts = false;
  pl->set_parity = p;
  pl->d_counts = pl->set_counts; pl->dflag = pl->set_dflag; pl->slow_items = pl->set_items; pl->any_deferred = pl->set_counts + 8;
  pl->n = n; pl->dim = 0; pl->kind = 1;
  return pl->set_tab[p];
}
static int setplan_build(tfra_sparse_plan* pl, size_t n, const int64_t* ids, hipStream_t s, bool counts) {
  SetPlanLaunch L;
  int rc = setplan_prepare(pl, n, s, counts, &L);
  if (rc) return rc;
  if (counts) setplan_kernel<true><<<L.blocks, SP_NT, 0, s>>>(n, (const i64*)ids, L.m2, L.cur, L.old, L.next_use_count);
  else setplan_kernel<false><<<L.blocks, SP_NT, 0, s>>>(n, (const i64*)ids, L.m2, L.cur, L.old, L.next_use_count);
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: launch failed");
  return TFRA_OK;
}

extern "C" int tfra_sparse_plan_build(tfra_sparse_plan_t* pl, size_t n, const int64_t* ids, int dim, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!pl) return set_error(TFRA_ERR_INVALID, "sparse_plan_build: null plan");
  const bool skip_counts = pl->skip_counts_once;   // consumed here on EVERY path (an empty batch or an error must not leak it into the next build)
  pl->skip_counts_once = false;
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != pl->device) { if (hipSetDevice(pl->device) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: hipSetDevice"); } }
  pl->n = 0;
  if (n == 0) return TFRA_OK;
  if (!ids) return set_error(TFRA_ERR_INVALID, "sparse_plan_build: null ids");
  if (dim == 0) return setplan_build(pl, n, ids, s, !skip_counts);   // assign-only: the last position and the count of every distinct id
  if (dim < 0 || dim % 4 != 0 || dim > 64 * MAXCH)
    return set_error(TFRA_ERR_UNSUPPORTED, "sparse_plan_build: needs dim % 4 == 0 and dim <= 256 (dim 0: assign-only plan)");
  if (n > MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "sparse_plan_build: at most 2^18 ids per plan");
  const size_t ntiles = (n + TILE - 1) / TILE, npad = ntiles * TILE;
  unsigned P = 64;
  while (P < 2048 && (size_t)P * 128 < n) P <<= 1;
  const unsigned cm = ntiles <= 256 ? 512u : 1024u;   // a key present in every tile must fit one pass with room to spare
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t reg = (size_t)P * CMAX;
  const size_t head = al((size_t)(P + 2) * CSTRIDE * 4);
  const size_t ctr = al((size_t)(NSH + 1) * CTR_STRIDE * 8);
  // per-shard capacities: twice the fair share + slack; bins / partial rows also cover ONE key holding every id
  CsrOut& o = pl->out;
  o.cr = (unsigned)(2 * npad / NSH + 64);
  o.hr = (unsigned)(2 * (npad / 9) / NSH + 64);
  o.pr = (unsigned)(npad / SEG + 2 * (npad / 9) / NSH + 64);
  o.br = (unsigned)(npad / SEG + (npad / 16 + npad / 9) / NSH / 16 + 2 * (P / NSH + 1) + 32);
  const size_t nrec_c = (size_t)NSH * o.cr, nrec_h = (size_t)NSH * o.hr, nbin = (size_t)NSH * o.br, npart = (size_t)NSH * o.pr;
  size_t bytes = head + ctr + 256                                               // cursors, counters, d_counts
                 + al(reg * 8) + al(reg * 4) + al(npad * 8) + 2 * al(npad * 4)  // descriptors + overflow list
                 + al(npad * 4) + al(npad * 2) + al(ntiles * 4) + al(npad * 16) // tile lists, run starts, lengths, destinations
                 + al(nrec_c * REC_WORDS * 4) + al(nrec_h * REC_WORDS * 4)      // key records
                 + al(nbin * SEG * 4) + al(nbin * 32 * 4)                       // bins + run outputs
                 + al(npad * 4) + al(npad * 8) + al(nbin * 4)                   // keymap, dense keys, binmap
                 + al(npad) + al((size_t)SLOW_CAP * sizeof(OwnItem))            // deferred flags, left-over item list
                 + al(npart * 4)                                                // partial row -> destination (route helpers)
                 + al(npart * (size_t)dim * 4);                                 // partial rows
  if (pl->bytes < bytes) {
    if (pl->buf) {
      if (hipDeviceSynchronize() != hipSuccess || hipFree(pl->buf) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: free");
      pl->buf = nullptr; pl->bytes = 0;
    }
    hipError_t e = hipMalloc(&pl->buf, bytes);
    if (e != hipSuccess) { pl->buf = nullptr; return set_error(e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP, "sparse_plan_build: hipMalloc"); }
    pl->bytes = bytes;
    pl->armed = false;
    pl->dflag = nullptr;
  }
  unsigned char* w = (unsigned char*)pl->buf;
  pl->cursors = (unsigned*)w; w += head;
  o.counters = (u64*)w; w += ctr;
  pl->d_counts = (unsigned*)w; w += 256;
  const bool same_layout = pl->armed && pl->P == P;
  if (!same_layout && hipMemsetAsync(pl->buf, 0, head + ctr + 256, s) != hipSuccess)
    return set_error(TFRA_ERR_HIP, "sparse_plan_build: memset");
  CsrDesc& ds = pl->ds;
  ds.key = (i64*)w; w += al(reg * 8);
  ds.ord = (unsigned*)w; w += al(reg * 4);
  ds.ovf_key = (i64*)w; w += al(npad * 8);
  ds.ovf_ord = (unsigned*)w; w += al(npad * 4);
  ds.ovf_bucket = (unsigned*)w; w += al(npad * 4);
  pl->tile_entries = (unsigned*)w; w += al(npad * 4);
  pl->run_start = (unsigned short*)w; w += al(npad * 2);
  pl->tile_len = (unsigned*)w; w += al(ntiles * 4);
  pl->drec = (uint4*)w; w += al(npad * 16);
  o.crec = (unsigned*)w; w += al(nrec_c * REC_WORDS * 4);
  o.hrec = (unsigned*)w; w += al(nrec_h * REC_WORDS * 4);
  o.hent = (unsigned*)w; w += al(nbin * SEG * 4);
  o.hout = (unsigned*)w; w += al(nbin * 32 * 4);
  pl->keymap = (unsigned*)w; w += al(npad * 4);
  pl->dkeys = (i64*)w; w += al(npad * 8);
  pl->binmap = (unsigned*)w; w += al(nbin * 4);
  if (pl->dflag != (uint8_t*)w || pl->dflag_len != npad) {   // the left-over flags are all zero between launches (upsert_own_kernel)
    if (hipMemsetAsync(w, 0, al(npad), s) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: memset");
    pl->dflag_len = npad;
  }
  pl->dflag = (uint8_t*)w; w += al(npad);
  pl->slow_items = (OwnItem*)w; w += al((size_t)SLOW_CAP * sizeof(OwnItem));
  pl->any_deferred = pl->d_counts + 8;
  pl->prow_dest = (int*)w; w += al(npart * 4);
  pl->partial = (float*)w;
  pl->gen += 1;
  ds.cursor = pl->cursors;
  ds.ovf_count = pl->cursors + (size_t)P * CSTRIDE;
  ds.ovf_cap = (unsigned)npad;
  unsigned* err = pl->cursors + (size_t)(P + 1) * CSTRIDE;
  csr_tile_kernel<<<dim3((unsigned)ntiles), NTA, 0, s>>>(n, (const i64*)ids, P, cm, ds, err, pl->tile_entries, pl->run_start, pl->tile_len, o.counters);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(csr_bucket_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)plan_smem_bytes(1024));
    attr_set = true;
  }
  csr_bucket_kernel<<<dim3(P), NT, plan_smem_bytes(cm), s>>>(P, (unsigned)ntiles, cm, ds, pl->drec, o, err);
  csr_scatter_kernel<<<dim3((unsigned)ntiles), NTA, 0, s>>>(pl->tile_entries, pl->run_start, pl->tile_len, pl->drec, o, pl->keymap,
                                                           pl->dkeys, pl->binmap, pl->cursors, P, pl->d_counts, pl->host_counts, pl->gen);
  pl->armed = true;
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_build: launch failed");
  pl->n = n; pl->npad = npad; pl->ntiles = ntiles; pl->P = P; pl->cm = cm; pl->dim = dim; pl->kind = 0;
  return TFRA_OK;
}

// the plan's counts as the host knows them: exact when the build has completed and published them
// (step driver), otherwise the worst case (grid-stride kernels read the real counts from device memory)
static void plan_grids(const tfra_sparse_plan* pl, unsigned* key_blocks, unsigned* bin_blocks) {
  size_t keys = pl->n, bins = (size_t)NSH * pl->out.br;
  if (pl->host_counts && (int)(*(volatile unsigned*)pl->host_counts - pl->gen) >= 0) {
    keys = (size_t)pl->host_counts[1] + pl->host_counts[2];
    bins = pl->host_counts[4];
  } else {
    keys = std::min<size_t>(keys, 32768);   // grid-stride beyond 2048 blocks
    bins = std::min<size_t>(bins, 1024);
  }
  *key_blocks = (unsigned)std::max<size_t>(1, (keys * 16 + 255) / 256);
  *bin_blocks = (unsigned)std::max<size_t>(1, bins);
}

static CsrKeys keys_of(const tfra_sparse_plan* pl) {
  if (pl->kind == 1) {
    const SetTab& tb = pl->set_tab[pl->set_parity];
    return CsrKeys{nullptr, nullptr, nullptr, nullptr, nullptr, tb.count - 1, tb.ukeys, tb.uslot, tb.ent};
  }
  return CsrKeys{pl->keymap, pl->dkeys, pl->out.crec, pl->out.hrec, pl->out.hent, pl->d_counts, nullptr, nullptr, nullptr};
}

template <int KIND>
static void launch_apply_csr(Table* t, hipStream_t s, const tfra_sparse_plan* pl, const OptP& o, const float* grads,
                             const float* default_row, unsigned key_blocks, const ScoreP& sp) {
  TableView v = t->view_of(t->cur);
  const float a0 = t->opts.aux_init[0], a1 = t->opts.aux_init[1];
  const unsigned gen = ++pl->use_gen;
  // one RESIDENT grid (the kernel is grid-stride: 125 registers = 4 blocks per CU x 256 CUs): a batch's 33 K keys as 2075 blocks were two
  // rounds of dispatch plus a third of 27 blocks; 1024 blocks looping twice: configs[1] 55.2 -> 53.7 us per step (A/B on one box, twice)
  static const unsigned grid_cap = [] { const char* e = getenv("TFRA_APPLY_GRID_CAP"); return e ? (unsigned)atoi(e) : 1024u; }();
  if (grid_cap) key_blocks = std::min(key_blocks, grid_cap);
  apply_csr_kernel<KIND, false><<<key_blocks, 256, 0, s>>>(v, o, pl->dim, grads, pl->partial, keys_of(pl), default_row, a0, a1, sp,
                                                           pl->dflag, pl->any_deferred, gen);
  if (sp.bounded)
    apply_csr_kernel<KIND, true><<<key_blocks, 256, 0, s>>>(v, o, pl->dim, grads, pl->partial, keys_of(pl), default_row, a0, a1, sp,
                                                            pl->dflag, pl->any_deferred, gen);
}

static int apply_planned_impl(tfra_table_t* tp, const tfra_opt_params* p, const tfra_sparse_plan_t* pl, const float* grads,
                              const float* param_default_row, tfra_stream_t stream, unsigned* progress, unsigned progress_val) {
  // caller holds t->mu
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !p || !pl) return set_error(TFRA_ERR_INVALID, "apply_planned: null argument");
  hipStream_t s = (hipStream_t)stream;
  int rc = t->enter(s);
  if (rc) return rc;
  if (pl->n == 0) return TFRA_OK;
  if (!grads || !param_default_row) return set_error(TFRA_ERR_INVALID, "apply_planned: null buffer");
  if (t->opts.value_dtype != TFRA_F32) return set_error(TFRA_ERR_UNSUPPORTED, "apply_planned: value_dtype must be float32");
  if (p->kind < 0 || p->kind > TFRA_OPT_FTRL) return set_error(TFRA_ERR_INVALID, "apply_planned: unknown kind");
  int need = p->kind == TFRA_OPT_SGD ? 0 : (p->kind == TFRA_OPT_ADAGRAD ? 1 : 2);
  if (t->opts.aux_fields < need) return set_error(TFRA_ERR_INVALID, "apply_planned: table lacks optimizer slot fields");
  if (t->opts.dim != pl->dim) return set_error(TFRA_ERR_INVALID, "apply_planned: the plan was built for another dim");
  if (t->opts.device != pl->device && t->opts.device >= 0) return set_error(TFRA_ERR_INVALID, "apply_planned: plan and table live on different devices");
  if ((((uintptr_t)grads | (uintptr_t)param_default_row) & 15))
    return set_error(TFRA_ERR_UNSUPPORTED, "apply_planned: gradient / default buffers must be 16-B aligned");
  rc = t->prepare_insert(pl->n, s);
  if (rc) return rc;
  const int dim = pl->dim;
  unsigned key_blocks, bin_blocks;
  plan_grids(pl, &key_blocks, &bin_blocks);
  const int nch = (dim + 63) / 64;
  switch (nch) {
    case 1: hot_sums_kernel<1><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, progress, progress_val); break;
    case 2: hot_sums_kernel<2><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, progress, progress_val); break;
    case 3: hot_sums_kernel<3><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, progress, progress_val); break;
    default: hot_sums_kernel<4><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, progress, progress_val); break;
  }
  uint8_t* bounded_now;
  rc = t->bounded_flags(1, s, &bounded_now);
  if (rc) return rc;
  const ScoreP sp{t->opts.strategy, t->global_epoch, bounded_now ? (t->dense ? 2 : 1) : 0};
  OptP o{p->kind, p->lr, p->beta1, p->beta2, p->eps, p->l1, p->l2, p->lr_power, p->d_lr};
  switch (p->kind) {
    case TFRA_OPT_SGD: launch_apply_csr<TFRA_OPT_SGD>(t, s, pl, o, grads, param_default_row, key_blocks, sp); break;
    case TFRA_OPT_ADAM: launch_apply_csr<TFRA_OPT_ADAM>(t, s, pl, o, grads, param_default_row, key_blocks, sp); break;
    case TFRA_OPT_ADAGRAD: launch_apply_csr<TFRA_OPT_ADAGRAD>(t, s, pl, o, grads, param_default_row, key_blocks, sp); break;
    default: launch_apply_csr<TFRA_OPT_FTRL>(t, s, pl, o, grads, param_default_row, key_blocks, sp); break;
  }
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "apply_planned: launch failed");
  step_epoch_public(t);
  return TFRA_OK;
}

extern "C" int tfra_table_apply_planned(tfra_table_t* tp, const tfra_opt_params* p, const tfra_sparse_plan_t* pl,
                                        const float* grads, const float* param_default_row, tfra_stream_t stream) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t) return set_error(TFRA_ERR_INVALID, "apply_planned: null table");
  std::lock_guard<std::mutex> lock(t->mu);
  return apply_planned_impl(tp, p, pl, grads, param_default_row, stream, nullptr, 0);
}

// ---- launch of the ownership write-back (plan keys or a caller's unique keys) --------------------------------
template <int SRC>
static void launch_own(hipStream_t s, int g, bool simple, const OwnArgs& a, size_t nkeys, OwnCtrs* ctr, OwnCtrs* next_ctr, unsigned og,
                       unsigned rest_blocks, unsigned* progress, unsigned progress_val) {
  // 4 waves x 16 keys per block and pass; up to a batch's worth of keys 8 keys per wave instead (own_batch16: U): twice the
  // waves, half the dependent work in each, 86 instead of 118 registers.  Measured on the 10^9-slot table: 22.7 K keys
  // 13.1 -> 11.3 us (the step 42.5 -> 41.1), 78 K keys the same kernel time alone and the step 64.0 -> 60.7 us; 4 keys per
  // wave: nothing more on 22.7 K keys (40.8 us), 64.6 us on 78 K
  const bool half = g == 16 && nkeys <= 131072;
  const unsigned blocks = (unsigned)std::max<size_t>(1, half ? (nkeys + 31) / 32 : (nkeys + 63) / 64);
  // a.tags == nullptr (TFRA_OPTION_NO_OWNER_TAGS, or the tags did not allocate): the locked protocol for every key
  const bool smp = a.stats_host != nullptr && (SRC == SRC_SET || SRC == SRC_DIRECT) && g == 16;   // somebody reads the sample
#define TFRA_OWN(GG, SS, UU)                                                                                  \
  if (a.tags) {                                                                                               \
    if (GG == 16 && smp) upsert_own_kernel<GG, SS, SRC, UU, false, false, GG == 16><<<blocks, 256, 0, s>>>(a, ctr, og, progress, progress_val); \
    else upsert_own_kernel<GG, SS, SRC, UU><<<blocks, 256, 0, s>>>(a, ctr, og, progress, progress_val);       \
    upsert_rest_kernel<GG, SRC><<<rest_blocks, 256, 0, s>>>(a, &ctr->n_a, reinterpret_cast<unsigned*>(next_ctr)); \
  } else {                                                                                                    \
    upsert_rest_kernel<GG, SRC><<<(unsigned)std::max<size_t>(1, (nkeys + 15) / 16), 256, 0, s>>>(a, nullptr, nullptr); \
  }
  // The form of the pass (16-byte granules): HF when the last write-back that has ended was mostly plain hits — fewer than a quarter of
  // the keys its sample looked at were new, evicting or left over — (TFRA_OWN_HF=1: always, 0: never; tuning, tests).  Where it may be
  // taken: over a SET plan's keys (a victim is checked against the plan), and over ANY keys — a caller's unique keys included: the
  // reference's Insert op — on a table that never evicts (unbounded: TFRA's default cuckoo flavour), where the only thing a hit's claim
  // protected it from does not exist.
  // (A caller's unique keys on a table that DOES evict have no plan to check a victim against.  The form in which every key in need of
  // a victim goes to the remainder — no eviction inside the pass, so a hit still needs no claim — was measured on the metric's table,
  // picked below one such key in 64: find + Insert of prepared keys 33.4 -> 29.4 us, but the Insert behind the lookup op 41.5 -> 43.1 us
  // and the pair alone 16.4 -> 19.1 us.  Not taken; TFRA_OWN_HF=1 still forces it, for the tests.)
  const char* hf_env = getenv("TFRA_OWN_HF");
  const bool hf_forced = hf_env && *hf_env && atoi(hf_env) != 0;
  if ((SRC == SRC_SET || SRC == SRC_DIRECT) && g == 16 && a.tags && (hf_forced || (SRC == SRC_SET && a.own_set.ent) || a.sp.bounded == 0)) {
    const char* e = hf_env;
    bool hf = false;
    if (e && *e) hf = atoi(e) != 0;
    else if (a.stats_host) {
      const unsigned nh = reinterpret_cast<const volatile unsigned*>(a.stats_host)[0], lk = reinterpret_cast<const volatile unsigned*>(a.stats_host)[1];
      hf = lk >= 32 && (size_t)nh * 4 < (size_t)lk;
    }
    if (hf) {
#define TFRA_OWN_HF_LAUNCH(SS, UU) upsert_own_kernel<16, SS, SRC, UU, false, true, true><<<blocks, 256, 0, s>>>(a, ctr, og, progress, progress_val)
      if (simple) { if (half) TFRA_OWN_HF_LAUNCH(true, 2); else TFRA_OWN_HF_LAUNCH(true, 4); }
      else { if (half) TFRA_OWN_HF_LAUNCH(false, 2); else TFRA_OWN_HF_LAUNCH(false, 4); }
#undef TFRA_OWN_HF_LAUNCH
      upsert_rest_kernel<16, SRC><<<rest_blocks, 256, 0, s>>>(a, &ctr->n_a, reinterpret_cast<unsigned*>(next_ctr));
      return;
    }
  }
  switch (g) {
    case 16:
      if (simple) { if (half) { TFRA_OWN(16, true, 2); } else { TFRA_OWN(16, true, 4); } }
      else { if (half) { TFRA_OWN(16, false, 2); } else { TFRA_OWN(16, false, 4); } }
      break;
    case 8: TFRA_OWN(8, false, 4); break;
    case 4: TFRA_OWN(4, false, 4); break;
    case 2: TFRA_OWN(2, false, 4); break;
    default: TFRA_OWN(1, false, 4); break;
  }
#undef TFRA_OWN
}

// the same pair as the reference's insert_or_accum (own_batch16: ACC) over a caller's unique keys; 16-byte granules only
static void launch_own_accum(hipStream_t s, bool simple, const OwnArgs& a, size_t nkeys, OwnCtrs* ctr, OwnCtrs* next_ctr, unsigned og,
                             unsigned rest_blocks) {
  const bool half = nkeys <= 131072;
  const unsigned blocks = (unsigned)std::max<size_t>(1, half ? (nkeys + 31) / 32 : (nkeys + 63) / 64);
#define TFRA_OWN_ACC(SS, UU) upsert_own_kernel<16, SS, SRC_DIRECT, UU, true><<<blocks, 256, 0, s>>>(a, ctr, og, nullptr, 0)
  if (simple) { if (half) TFRA_OWN_ACC(true, 2); else TFRA_OWN_ACC(true, 4); }
  else { if (half) TFRA_OWN_ACC(false, 2); else TFRA_OWN_ACC(false, 4); }
#undef TFRA_OWN_ACC
  upsert_rest_kernel<16, SRC_DIRECT, true><<<rest_blocks, 256, 0, s>>>(a, &ctr->n_a, reinterpret_cast<unsigned*>(next_ctr));
}

// Expected left-over keys of an ownership pass over `nkeys` keys: two keys sharing a home bucket, (2 n)^2 / (2 nb).
static double expect_leftover(double nkeys, double nb) { return 2.0 * nkeys * nkeys / nb; }

static unsigned next_own_gen(Table* t) {
  if (++t->own_gen == 0) t->own_gen = 1;     // bucket-owner tag of this launch (tags start at 0; a stale equal tag after a wrap only
  return t->own_gen;                         // sends a key to the remainder pass)
}

// Host half of an ownership write-back of a plan's keys: everything but the launches (upsert_planned_impl launches the pair
// upsert_own_kernel + upsert_rest_kernel, the overlapped step puts the pass into its one launch).
struct OwnLaunch {
  OwnArgs a;
  OwnCtrs* ctr; OwnCtrs* next_ctr;
  unsigned og;            // bucket-owner generation of this launch (0: no owner tags)
  unsigned key_blocks;    // the plan's keys / 16, as far as the host knows them
  unsigned rem_blocks;    // grid of the remainder pass
  bool simple;
  int g;                  // copy granule
};
static int own_prepare(Table* t, const tfra_sparse_plan_t* pl, const void* values, const uint64_t* scores, hipStream_t s,
                       const unsigned* progress, OwnLaunch* L) {
  // caller holds t->mu and has called t->enter(s)
  if (!values) return set_error(TFRA_ERR_INVALID, "upsert_planned: null values");
  if (t->opts.device != pl->device && t->opts.device >= 0) return set_error(TFRA_ERR_INVALID, "upsert_planned: plan and table live on different devices");
  if (pl->kind == 1 && !pl->built_counts && t->opts.strategy == TFRA_EVICT_LFU && !scores)
    return set_error(TFRA_ERR_INVALID, "upsert_planned: this plan was built without occurrence counts (by a step driver of a table whose scores do not read them); an LFU table without caller scores needs them");
  int rc = t->prepare_insert(pl->n, s);
  if (rc) return rc;
  unsigned key_blocks, bin_blocks;
  plan_grids(pl, &key_blocks, &bin_blocks);
  if (pl->kind == 1 && progress) {
    // an assign-only plan does not tell the host how many distinct keys it found; the step driver's batches resemble each
    // other: the count the last write-back that has started saw, plus a quarter (a batch with more: grid-stride)
    const unsigned seen = reinterpret_cast<const volatile unsigned*>(progress)[1];
    if (seen) key_blocks = (unsigned)std::max<size_t>(1, (std::min<size_t>(pl->n, (size_t)seen + seen / 4 + 1024) * 16 + 255) / 256);
  }
  uint8_t* bounded_now;
  rc = t->bounded_flags(1, s, &bounded_now);
  if (rc) return rc;
  const ScoreP sp{t->opts.strategy, t->global_epoch, bounded_now ? (t->dense ? 2 : 1) : 0};
  size_t x = (size_t)t->field_bytes | (size_t)(uintptr_t)values | 16;
  int g = (int)(x & (~x + 1));
  if (g > 16) g = 16;
  ++pl->use_gen;
  unsigned* tags = t->ensure_own_tags(s);    // nullptr (no owner tags): every key takes the locked protocol
  L->og = tags ? next_own_gen(t) : 0;
  const unsigned par = pl->ups_uses[pl->kind == 1 ? 1 : 0]++ & 1u;   // (its own count per buffer: apply_planned uses of the plan do not touch the counters)
  L->ctr = reinterpret_cast<OwnCtrs*>(pl->d_counts + 12) + par;
  L->next_ctr = reinterpret_cast<OwnCtrs*>(pl->d_counts + 12) + (par ^ 1u);
  // Left-over keys of the ownership pass.  Few (a big table): the remainder kernel walks their list with a handful of
  // blocks.  Many (a small table): full grid.
  const double nkeys = (double)key_blocks * 16.0;   // unique keys of the plan when its counts have arrived, else the id count
  L->rem_blocks = expect_leftover(nkeys, (double)t->cur.nb) < 2048.0 ? 32u : key_blocks;
  L->key_blocks = key_blocks;
  L->simple = t->opts.aux_fields == 0 && t->opts.strategy == TFRA_EVICT_LRU && !scores;
  L->g = g;
  OwnArgs& a = L->a;
  a = OwnArgs{};
  a.v = t->view_of(t->cur); a.vals = (const unsigned char*)values; a.scores = (const u64*)scores; a.ks = keys_of(pl); a.keys = nullptr; a.nkeys = 0;
  a.ai = t->aux; a.sp = sp;
  a.dflag = pl->dflag; a.tags = tags; a.items = pl->slow_items; a.item_cap = SLOW_CAP;
  if (pl->kind == 1) {   // a SET plan: its table as something to probe (HF), and where the pass's sample goes
    const SetTab& tb = pl->set_tab[pl->set_parity];
    a.own_set = SetProbe{tb.ent, pl->set_m2};
    a.stats_host = t->own_stats_host;
  }
  return TFRA_OK;
}

static int upsert_planned_impl(tfra_table_t* tp, const tfra_sparse_plan_t* pl, const void* values, const uint64_t* scores,
                               tfra_stream_t stream, unsigned* progress, unsigned progress_val) {
  // caller holds t->mu
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !pl) return set_error(TFRA_ERR_INVALID, "upsert_planned: null argument");
  hipStream_t s = (hipStream_t)stream;
  int rc = t->enter(s);
  if (rc) return rc;
  if (pl->n == 0) return TFRA_OK;
  OwnLaunch L;
  rc = own_prepare(t, pl, values, scores, s, progress, &L);
  if (rc) return rc;
  if (pl->kind == 1) launch_own<SRC_SET>(s, L.g, L.simple, L.a, (size_t)L.key_blocks * 16, L.ctr, L.next_ctr, L.og, L.rem_blocks, progress, progress_val);
  else launch_own<SRC_PLAN>(s, L.g, L.simple, L.a, (size_t)L.key_blocks * 16, L.ctr, L.next_ctr, L.og, L.rem_blocks, progress, progress_val);
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "upsert_planned: launch failed");
  step_epoch_public(t);
  return TFRA_OK;
}

// tfra_table_insert_or_assign with TFRA_FLAG_UNIQUE_KEYS (the reference's Insert op hands HKV unique keys,
// hkv_hashtable_op_gpu.cu.cc:253-290 -> lookup_table_op_hkv.h:522-537): the same single pass with bucket ownership, fed
// with the caller's key array — value row i belongs to key i, no plan.  *taken = false: not for this call (no owner tags,
// or so many keys for the table's size that most of them would collide on a home bucket: a bulk load) — the caller runs
// the locked two-phase kernels.  Caller holds t->mu and has called prepare_insert.
namespace tfra {
int own_upsert_unique(Table* t, hipStream_t s, size_t n, const i64* keys, const void* values, const u64* scores, bool* taken,
                      const uint8_t* accum_exists, const int64_t* d_n) {
  // accum_exists != nullptr: insert_or_accum (tfra_table_accum_or_assign with TFRA_FLAG_UNIQUE_KEYS) instead of an assign
  *taken = false;
  if (accum_exists && (((size_t)t->field_bytes | (size_t)(uintptr_t)values) & 15)) return TFRA_OK;   // 16-byte granules only
  if (n == 0 || n > (1u << 24)) return TFRA_OK;
  const double expect = expect_leftover((double)n, (double)t->cur.nb);
  if (expect >= 2048.0) return TFRA_OK;   // most keys would collide on a home bucket (a bulk load): the locked kernels
  // the table's own scratch of this path: 2 counter sets | item list | one flag byte per key
  const size_t head = 256 + (size_t)SLOW_CAP * sizeof(OwnItem);
  const size_t need = head + ((n + 255) / 256) * 256;
  if (t->capture_safe && (t->own_ws_bytes < need || !t->own_tags || t->own_tags_nb != t->cur.nb)) return TFRA_OK;   // no allocation while capturing
  unsigned* tags = t->ensure_own_tags(s);
  if (!tags) return TFRA_OK;
  if (t->own_ws_bytes < need) {
    if (t->own_ws) { if (hipStreamSynchronize(s) != hipSuccess) return set_error(TFRA_ERR_HIP, "insert: sync"); t->dfree(t->own_ws, s); t->own_ws = nullptr; t->own_ws_bytes = 0; }
    const size_t want = head + std::max<size_t>(((n + 255) / 256) * 256, (size_t)1 << 18);
    t->own_ws = t->dalloc(want, s);
    if (!t->own_ws) { g_last_error.clear(); return TFRA_OK; }   // no scratch: the locked kernels need none
    if (hipMemsetAsync(t->own_ws, 0, want, s) != hipSuccess) return set_error(TFRA_ERR_HIP, "insert: memset");   // counters and flags start at zero
    t->own_ws_bytes = want;
    t->own_ws_uses = 0;
  }
  uint8_t* bounded_now;
  int rc = t->bounded_flags(1, s, &bounded_now);
  if (rc) return rc;
  const ScoreP sp{t->opts.strategy, t->global_epoch, bounded_now ? (t->dense ? 2 : 1) : 0};
  size_t x = (size_t)t->field_bytes | (size_t)(uintptr_t)values | 16;
  int g = (int)(x & (~x + 1));
  if (g > 16) g = 16;
  const unsigned og = next_own_gen(t);
  const unsigned par = t->own_ws_uses++ & 1u;
  OwnCtrs* ctr = reinterpret_cast<OwnCtrs*>(t->own_ws) + par;
  OwnCtrs* next_ctr = reinterpret_cast<OwnCtrs*>(t->own_ws) + (par ^ 1u);
  const bool simple = t->opts.aux_fields == 0 && t->opts.strategy == TFRA_EVICT_LRU && !scores;
  OwnArgs a{};
  a.v = t->view_of(t->cur); a.vals = (const unsigned char*)values; a.scores = scores; a.keys = keys; a.nkeys = (unsigned)n;
  a.ai = t->aux; a.sp = sp; a.dflag = (uint8_t*)t->own_ws + head; a.tags = tags;
  a.items = reinterpret_cast<OwnItem*>((unsigned char*)t->own_ws + 256); a.item_cap = SLOW_CAP;
  a.exists = accum_exists; a.acc_dt = t->opts.value_dtype; a.d_nkeys = (const long long*)d_n;
  a.stats_host = sp.bounded == 0 ? t->own_stats_host : nullptr;   // (a table that evicts never takes the other form for a caller's keys: nothing to sample for)
  if (accum_exists) launch_own_accum(s, simple, a, n, ctr, next_ctr, og, 32u);
  else launch_own<SRC_DIRECT>(s, g, simple, a, n, ctr, next_ctr, og, 32u, nullptr, 0);
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "insert: launch failed");
  *taken = true;
  return TFRA_OK;
}
}  // namespace tfra

extern "C" int tfra_table_upsert_planned(tfra_table_t* tp, const tfra_sparse_plan_t* pl, const void* values,
                                         const uint64_t* scores, tfra_stream_t stream) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t) return set_error(TFRA_ERR_INVALID, "upsert_planned: null table");
  std::lock_guard<std::mutex> lock(t->mu);
  return upsert_planned_impl(tp, pl, values, scores, stream, nullptr, 0);
}

// Introspection for tests and tools: the plan's CSR as flat arrays (copies; synchronises the stream).
//   counts[6]; keys[nkeys] (keys with many occurrences first); cnt[nkeys]; positions[n]: key 0's, key 1's, ... each ascending
extern "C" int tfra_sparse_plan_read(const tfra_sparse_plan_t* pl, uint32_t* counts, int64_t* keys, uint32_t* cnt,
                                     uint32_t* positions, size_t cap, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!pl || !counts) return set_error(TFRA_ERR_INVALID, "sparse_plan_read: null argument");
  if (pl->n == 0) { for (int i = 0; i < 6; ++i) counts[i] = 0; return TFRA_OK; }
  if (hipMemcpyAsync(counts, keys_of(pl).d_counts, 6 * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return set_error(TFRA_ERR_HIP, "sparse_plan_read: copy");
  const unsigned nhot = counts[0], ncold = counts[1], nbins = counts[3];
  if (!keys) return TFRA_OK;
  if (pl->kind == 1 && pl->tab_state[pl->set_parity] == TAB_LISTLESS)
    return set_error(TFRA_ERR_UNSUPPORTED, "sparse_plan_read: this plan was built by the overlapped step, without a key list");
  if (pl->kind == 1) {   // SET plan: distinct keys and their occurrence counts; it keeps no positions list
    if (positions) return set_error(TFRA_ERR_UNSUPPORTED, "sparse_plan_read: an assign-only plan keeps the last position of a key, not the list of its positions");
    if ((size_t)ncold > cap) return set_error(TFRA_ERR_INVALID, "sparse_plan_read: buffers too small");
    const SetTab& tb = pl->set_tab[pl->set_parity];
    std::vector<unsigned> sl(ncold);
    std::vector<SetEnt> pc((size_t)pl->set_m2 + 2);
    if ((ncold && hipMemcpyAsync(keys, tb.ukeys, (size_t)ncold * 8, hipMemcpyDeviceToHost, s) != hipSuccess) ||
        (ncold && hipMemcpyAsync(sl.data(), tb.uslot, (size_t)ncold * 4, hipMemcpyDeviceToHost, s) != hipSuccess) ||
        hipMemcpyAsync(pc.data(), tb.ent, pc.size() * sizeof(SetEnt), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return set_error(TFRA_ERR_HIP, "sparse_plan_read: copy");
    for (unsigned i = 0; i < ncold; ++i) cnt[i] = pc[sl[i]].cnt;
    return TFRA_OK;
  }
  if ((size_t)nhot + ncold > cap) return set_error(TFRA_ERR_INVALID, "sparse_plan_read: buffers too small");
  const CsrOut& o = pl->out;
  const size_t nrec_c = (size_t)NSH * o.cr, nrec_h = (size_t)NSH * o.hr, nbin = (size_t)NSH * o.br;
  std::vector<unsigned> km((size_t)nhot + ncold), bm(nbins), crec(nrec_c * REC_WORDS), hrec(nrec_h * REC_WORDS), hent(nbin * SEG), hout(nbin * 32);
  auto cp = [&](void* d, const void* sdev, size_t b) { return b == 0 || hipMemcpyAsync(d, sdev, b, hipMemcpyDeviceToHost, s) == hipSuccess; };
  bool ok = cp(km.data(), pl->keymap, km.size() * 4) && cp(bm.data(), pl->binmap, bm.size() * 4) && cp(crec.data(), o.crec, crec.size() * 4) &&
            cp(hrec.data(), o.hrec, hrec.size() * 4) && cp(hent.data(), o.hent, hent.size() * 4) && cp(hout.data(), o.hout, hout.size() * 4);
  if (!ok || hipStreamSynchronize(s) != hipSuccess) return set_error(TFRA_ERR_HIP, "sparse_plan_read: copy");
  // partial row -> entry address of the head of its run (only the bins of this build)
  std::vector<std::pair<unsigned, size_t>> heads;
  for (unsigned i = 0; i < nbins; ++i) {
    const size_t bin = bm[i];
    for (unsigned it = 0; it < 32; ++it) {
      const unsigned e = hent[bin * SEG + it * 16];
      if ((e & E_HEAD) && !(e & E_SKIP)) heads.push_back({hout[bin * 32 + it], bin * SEG + it * 16});
    }
  }
  std::sort(heads.begin(), heads.end());
  size_t w = 0;
  for (size_t i = 0; i < km.size(); ++i) {
    const bool many = (km[i] & KM_MANY) != 0;
    const unsigned* rec = (many ? hrec.data() : crec.data()) + (size_t)(km[i] & ~KM_MANY) * REC_WORDS;
    keys[i] = (int64_t)(((uint64_t)rec[1] << 32) | rec[0]);
    cnt[i] = rec[2];
    if (!many) {
      for (unsigned j = 0; j < rec[2]; ++j) { if (w >= pl->n) return set_error(TFRA_ERR_INVALID, "sparse_plan_read: too many entries"); positions[w++] = rec[4 + j]; }
      continue;
    }
    for (unsigned pth = 0; pth < rec[4]; ++pth) {
      auto itp = std::lower_bound(heads.begin(), heads.end(), std::make_pair(rec[3] + pth, (size_t)0));
      if (itp == heads.end() || itp->first != rec[3] + pth) return set_error(TFRA_ERR_INVALID, "sparse_plan_read: a partial has no run");
      size_t a = itp->second;
      bool first = true;
      for (; (a % SEG != 0 || first) && (first || !(hent[a] & E_HEAD)); ++a) {
        first = false;
        if (!(hent[a] & E_SKIP)) { if (w >= pl->n) return set_error(TFRA_ERR_INVALID, "sparse_plan_read: too many entries"); positions[w++] = hent[a] & E_POS; }
      }
    }
  }
  return TFRA_OK;
}

// ---------------------------------------------------------------------------------------------
// One-call forms: the plan is built on the caller's stream into a plan object owned by the table.
static int own_plan(Table* t, tfra_sparse_plan** out) {
  if (!t->own_plan) {
    tfra_sparse_plan* pl = nullptr;
    int rc = tfra_sparse_plan_create(t->device, &pl);
    if (rc) return rc;
    t->own_plan = pl;
  }
  *out = reinterpret_cast<tfra_sparse_plan*>(t->own_plan);
  return TFRA_OK;
}

namespace tfra {
void destroy_own_plan(Table* t) {
  if (t->big_ws) { tfra_workspace_destroy(reinterpret_cast<tfra_workspace_t*>(t->big_ws)); t->big_ws = nullptr; }
  if (t->own_plan) { tfra_sparse_plan_destroy(reinterpret_cast<tfra_sparse_plan*>(t->own_plan)); t->own_plan = nullptr; }
}
}  // namespace tfra

// tfra_table_apply_sparse for more ids than a plan holds (2^18).  Equal ids must still meet in ONE update, whatever
// chunk they sit in:
//   1. per chunk of 2^18 ids: unique + per-key gradient sums (tfra_reduce_by_key) into one concatenated list — a key now
//      occurs at most once per chunk, so even the hottest id of a Zipf batch is a handful of entries;
//   2. the list fits a plan: one planned write-back sums a key's entries in chunk order and applies it;
//      else the list is split by key hash (tfra_partition, mode 2) into parts that fit — a key's entries stay together,
//      the parts are disjoint key sets — and each part is written back on its own.
// A slow path (host reads of the counts, scratch allocated per call); results are deterministic, the association of the
// sums is (within chunk) + (across chunks in order).
static int apply_sparse_big(Table* t, tfra_table_t* tp, tfra_sparse_plan* pl, const tfra_opt_params* p, size_t n, const int64_t* ids,
                            const float* grads, const float* param_default_row, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int dim = t->opts.dim;
  if (!t->big_ws) {
    tfra_workspace_t* w = nullptr;
    int rc = tfra_workspace_create(t->device, &w);
    if (rc) return rc;
    t->big_ws = w;
  }
  tfra_workspace_t* ws = reinterpret_cast<tfra_workspace_t*>(t->big_ws);
  i64 *keys_cat = nullptr, *d_cnt = nullptr, *keys_part = nullptr, *d_counts = nullptr;
  float *sums_cat = nullptr, *sums_part = nullptr;
  int* perm = nullptr;
  auto cleanup = [&](int rc) {
    (void)hipStreamSynchronize(s);
    (void)hipFree(keys_cat); (void)hipFree(d_cnt); (void)hipFree(keys_part); (void)hipFree(d_counts); (void)hipFree(sums_cat);
    (void)hipFree(sums_part); (void)hipFree(perm);
    return rc;
  };
  auto oom = [&]() { return cleanup(set_error(TFRA_ERR_OOM, "apply_sparse: scratch for a batch of more than 2^18 ids")); };
  if (hipMalloc((void**)&keys_cat, n * sizeof(i64)) != hipSuccess || hipMalloc((void**)&sums_cat, n * (size_t)dim * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&d_cnt, sizeof(i64)) != hipSuccess)
    return oom();
  size_t T = 0;
  for (size_t off = 0; off < n; off += MAX_IDS) {
    const size_t m = std::min<size_t>(MAX_IDS, n - off);
    int rc = tfra_reduce_by_key(ws, m, ids + off, dim, grads + off * (size_t)dim, (int64_t*)keys_cat + T, sums_cat + T * (size_t)dim,
                                (int64_t*)d_cnt, stream);
    if (rc) return cleanup(rc);
    i64 c = 0;
    if (hipMemcpyAsync(&c, d_cnt, sizeof(i64), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return cleanup(set_error(TFRA_ERR_HIP, "apply_sparse: count read"));
    if (c < 0) return cleanup(set_error(TFRA_ERR_FULL, "apply_sparse: a de-duplication plan overflowed"));
    T += (size_t)c;
  }
  if (T <= MAX_IDS) {
    int rc = tfra_sparse_plan_build(pl, T, (const int64_t*)keys_cat, dim, stream);
    if (!rc) rc = apply_planned_impl(tp, p, pl, sums_cat, param_default_row, stream, nullptr, 0);
    return cleanup(rc);
  }
  if (hipMalloc((void**)&keys_part, T * sizeof(i64)) != hipSuccess || hipMalloc((void**)&perm, T * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&sums_part, T * (size_t)dim * sizeof(float)) != hipSuccess)
    return oom();
  for (size_t P = (T + (MAX_IDS / 2) - 1) / (MAX_IDS / 2); P <= 2048; P *= 2) {
    (void)hipFree(d_counts); d_counts = nullptr;
    if (hipMalloc((void**)&d_counts, P * sizeof(i64)) != hipSuccess) return oom();
    int rc = tfra_partition(ws, T, nullptr, (const int64_t*)keys_cat, (int)P, 2, (int64_t*)keys_part, perm, (int64_t*)d_counts, stream);
    if (rc) return cleanup(rc);
    std::vector<i64> counts(P);
    if (hipMemcpyAsync(counts.data(), d_counts, P * sizeof(i64), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return cleanup(set_error(TFRA_ERR_HIP, "apply_sparse: count read"));
    bool fits = true;
    for (i64 c : counts) fits = fits && (size_t)c <= MAX_IDS;
    if (!fits) continue;   // a part too large (skewed hash): more parts
    rc = tfra_gather_rows(T, (size_t)dim * sizeof(float), sums_cat, perm, sums_part, stream);
    if (rc) return cleanup(rc);
    size_t off = 0;
    // ONE logical write-back: the epoch / step counters of the EPOCH* strategies advance once, not once per part (the
    // reference counts one upsert per write-back, lookup_table_op_hkv.h:528-536).  (On a bounded table at max_capacity a
    // later part may still evict keys an earlier part of the same batch inserted.)
    t->epoch_hold = true;
    for (i64 c : counts) {
      if (c > 0) {
        rc = tfra_sparse_plan_build(pl, (size_t)c, (const int64_t*)keys_part + off, dim, stream);
        if (!rc) rc = apply_planned_impl(tp, p, pl, sums_part + off * (size_t)dim, param_default_row, stream, nullptr, 0);
        if (rc) { t->epoch_hold = false; return cleanup(rc); }
      }
      off += (size_t)c;
    }
    t->epoch_hold = false;
    step_epoch_public(t);
    return cleanup(TFRA_OK);
  }
  return cleanup(set_error(TFRA_ERR_UNSUPPORTED, "apply_sparse: could not split the batch into parts of 2^18 ids"));
}

extern "C" int tfra_table_apply_sparse(tfra_table_t* tp, const tfra_opt_params* p, size_t n, const int64_t* ids,
                                       const float* grads, const float* param_default_row, tfra_stream_t stream) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !p) return set_error(TFRA_ERR_INVALID, "apply_sparse: null argument");
  if (n == 0) return TFRA_OK;
  if (!ids || !grads || !param_default_row) return set_error(TFRA_ERR_INVALID, "apply_sparse: null buffer");
  if (t->opts.value_dtype != TFRA_F32) return set_error(TFRA_ERR_UNSUPPORTED, "apply_sparse: value_dtype must be float32");
  const int dim = t->opts.dim;
  if (dim % 4 != 0 || dim > 64 * MAXCH || (((uintptr_t)grads | (uintptr_t)param_default_row) & 15))
    return set_error(TFRA_ERR_UNSUPPORTED, "apply_sparse: needs dim % 4 == 0, dim <= 256 and 16-B aligned buffers "
                                           "(use tfra_unique + tfra_segment_sum + tfra_table_apply_optimizer otherwise)");
  tfra_sparse_plan* pl;
  std::lock_guard<std::mutex> lock(t->mu);
  int rc = t->enter((hipStream_t)stream);   // orders the rebuild of the table's own plan behind its previous use
  if (rc) return rc;
  rc = own_plan(t, &pl);
  if (rc) return rc;
  if (n > MAX_IDS) return apply_sparse_big(t, tp, pl, p, n, ids, grads, param_default_row, stream);
  rc = tfra_sparse_plan_build(pl, n, ids, dim, stream);
  if (rc) return rc;
  return apply_planned_impl(tp, p, pl, grads, param_default_row, stream, nullptr, 0);
}

// insert_or_assign of a batch whose keys may repeat, last occurrence wins, dedup on the device
extern "C" int tfra_table_upsert_sparse(tfra_table_t* tp, size_t n, const int64_t* ids, const void* values,
                                        const uint64_t* scores, tfra_stream_t stream) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t) return set_error(TFRA_ERR_INVALID, "upsert_sparse: null table");
  if (n == 0) return TFRA_OK;
  if (!ids || !values) return set_error(TFRA_ERR_INVALID, "upsert_sparse: null buffer");
  tfra_sparse_plan* pl;
  std::lock_guard<std::mutex> lock(t->mu);
  int rc = t->enter((hipStream_t)stream);
  if (rc) return rc;
  rc = own_plan(t, &pl);
  if (rc) return rc;
  // more ids than a plan holds: chunk after chunk on the stream — a later chunk overwrites an earlier one, which is
  // "the last occurrence wins" across chunks too
  for (size_t off = 0; off < n; off += MAX_IDS) {
    const size_t m = std::min<size_t>(MAX_IDS, n - off);
    pl->skip_counts_once = !(t->opts.strategy == TFRA_EVICT_LFU && !scores);   // what reads a key's occurrence count (own_batch16)
    rc = tfra_sparse_plan_build(pl, m, ids + off, 0, stream);
    if (rc) return rc;
    rc = upsert_planned_impl(tp, pl, (const unsigned char*)values + off * (size_t)t->field_bytes, scores ? scores + off : nullptr, stream,
                             nullptr, 0);
    if (rc) return rc;
  }
  return TFRA_OK;
}

// unique + unsorted_segment_sum in one call = the plan + the hot sums + a gather (the reduction half of
// tfra_table_apply_sparse with the same summation tree, so routing the sums elsewhere — multi-GPU gradient
// alltoall — and applying them there gives the same bits as applying them here).
extern "C" int tfra_reduce_by_key(tfra_workspace_t* ws, size_t n, const int64_t* ids, int dim, const float* grads,
                                  int64_t* keys_out, float* rows_out, int64_t* d_count, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ws || !d_count) return set_error(TFRA_ERR_INVALID, "reduce_by_key: null argument");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != ws->device) { if (hipSetDevice(ws->device) != hipSuccess) return set_error(TFRA_ERR_HIP, "reduce_by_key: hipSetDevice"); } }
  if (n == 0) {
    if (hipMemsetAsync(d_count, 0, sizeof(int64_t), s) != hipSuccess) return set_error(TFRA_ERR_HIP, "reduce_by_key: memset");
    return TFRA_OK;
  }
  if (!ids || !grads || !keys_out || !rows_out) return set_error(TFRA_ERR_INVALID, "reduce_by_key: null buffer");
  if (dim <= 0 || dim % 4 != 0 || dim > 64 * MAXCH || (((uintptr_t)grads | (uintptr_t)rows_out) & 15))
    return set_error(TFRA_ERR_UNSUPPORTED, "reduce_by_key: needs dim % 4 == 0, dim <= 256 and 16-B aligned buffers "
                                           "(use tfra_unique + tfra_segment_sum otherwise)");
  if (n > MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "reduce_by_key: at most 2^18 ids per call");
  if (!ws->plan) {
    tfra_sparse_plan* np_ = nullptr;
    int rc = tfra_sparse_plan_create(ws->device, &np_);
    if (rc) return rc;
    ws->plan = np_;
  }
  tfra_sparse_plan* pl = reinterpret_cast<tfra_sparse_plan*>(ws->plan);
  int rc = tfra_sparse_plan_build(pl, n, ids, dim, stream);
  if (rc) return rc;
  unsigned key_blocks, bin_blocks;
  plan_grids(pl, &key_blocks, &bin_blocks);
  const int nch = (dim + 63) / 64;
  switch (nch) {
    case 1: hot_sums_kernel<1><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, nullptr, 0); break;
    case 2: hot_sums_kernel<2><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, nullptr, 0); break;
    case 3: hot_sums_kernel<3><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, nullptr, 0); break;
    default: hot_sums_kernel<4><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, nullptr, 0); break;
  }
  gather_csr_kernel<<<key_blocks, 256, 0, s>>>(dim, grads, pl->partial, keys_of(pl), (i64*)keys_out, rows_out, (i64*)d_count, nullptr);
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "reduce_by_key: launch failed");
  return TFRA_OK;
}

// The per-key gradient sums of a batch whose plan was built ahead (tfra_sparse_plan_build with the table's dim, on any
// stream): hot sums + gather, the reduction half of tfra_reduce_by_key with the same summation tree, written to
// rows_out[dest[p]] where p is a position of the key.  dest [n] int32: the caller's map from batch positions to output
// rows, the same for every position of a key (e.g. position -> owner-major index of the multi-GPU gradient route, which
// is known from the ids alone).  rows_out must have a row for every value in dest; rows no key maps to are not written.
extern "C" int tfra_plan_reduce_to(const tfra_sparse_plan_t* pl, const float* grads, const int32_t* dest, float* rows_out,
                                   tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!pl) return set_error(TFRA_ERR_INVALID, "plan_reduce_to: null plan");
  if (pl->kind == 1) return set_error(TFRA_ERR_UNSUPPORTED, "plan_reduce_to: needs a plan built with the table's dim");
  if (pl->n == 0) return TFRA_OK;
  if (!grads || !dest || !rows_out) return set_error(TFRA_ERR_INVALID, "plan_reduce_to: null buffer");
  const int dim = pl->dim;
  if (dim <= 0 || (((uintptr_t)grads | (uintptr_t)rows_out) & 15))
    return set_error(TFRA_ERR_INVALID, "plan_reduce_to: the plan must have been built with the rows' dim; buffers 16-B aligned");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != pl->device) { if (hipSetDevice(pl->device) != hipSuccess) return set_error(TFRA_ERR_HIP, "plan_reduce_to: hipSetDevice"); } }
  unsigned key_blocks, bin_blocks;
  plan_grids(pl, &key_blocks, &bin_blocks);
  const int nch = (dim + 63) / 64;
  switch (nch) {
    case 1: hot_sums_kernel<1><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, nullptr, 0); break;
    case 2: hot_sums_kernel<2><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, nullptr, 0); break;
    case 3: hot_sums_kernel<3><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, nullptr, 0); break;
    default: hot_sums_kernel<4><<<bin_blocks, NTA, 0, s>>>(grads, dim, pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->partial, nullptr, 0); break;
  }
  gather_csr_kernel<<<key_blocks, 256, 0, s>>>(dim, grads, pl->partial, keys_of(pl), nullptr, rows_out, nullptr, (const int*)dest);
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "plan_reduce_to: launch failed");
  return TFRA_OK;
}

// ---------------------------------------------------------------------------------------------
// Route helpers: the distinct keys of a built plan stand in for tf.unique (PY/shadow_embedding_ops.py:316 does unique,
// then partitions the distinct ids by owner).  tfra_plan_partition = tfra_partition over the plan's keys;
// tfra_plan_positions_to expands its perm to every batch position: dest[p] = j for the positions p of key perm[j].
__global__ __launch_bounds__(256) void plan_dest_keys_kernel(CsrKeys ks, const int* __restrict__ perm, int* __restrict__ dest,
                                                             int* __restrict__ prow_dest) {
  const unsigned total = ks.d_counts[0] + ks.d_counts[1];
  for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
    const unsigned km = ks.keymap[perm[j]];
    if (km & KM_MANY) {   // its positions sit in the bins: tag the key's partial rows, plan_dest_bins_kernel does the rest
      const unsigned* rec = ks.hrec + (size_t)(km & ~KM_MANY) * REC_WORDS;
      const unsigned first = rec[3], nsrc = rec[4];
      for (unsigned t = 0; t < nsrc; ++t) prow_dest[first + t] = (int)j;
    } else {
      const unsigned* rec = ks.crec + (size_t)km * REC_WORDS;
      const unsigned cnt = rec[2];
      for (unsigned e = 0; e < cnt; ++e) dest[rec[4 + e]] = (int)j;
    }
  }
}

// one block per bin of 512 entries, a 16-lane group per item (the layout hot_sums_kernel reduces)
__global__ __launch_bounds__(NTA) void plan_dest_bins_kernel(const unsigned* __restrict__ hent, const unsigned* __restrict__ hout,
                                                             const unsigned* __restrict__ binmap, const unsigned* __restrict__ d_counts,
                                                             const int* __restrict__ prow_dest, int* __restrict__ dest) {
  constexpr int NG = NTA / 16;
  __shared__ unsigned char s_kind[NG];   // 0 = continues the run of the item before, 1 = first item of a run, 2 = empty item
  __shared__ unsigned s_row[NG];
  const int lane = threadIdx.x & 63, sub = lane & 15, gshift = lane & 48, g = threadIdx.x >> 4;
  const unsigned nbins = d_counts[3];
  for (unsigned ib = blockIdx.x; ib < nbins; ib += gridDim.x) {
    const unsigned bin = binmap[ib];
    const unsigned e = hent[(size_t)bin * SEG + threadIdx.x];
    const unsigned e0 = (unsigned)__shfl((int)e, gshift);
    const bool empty = (e0 & E_SKIP) != 0, first = (e0 & E_HEAD) != 0;
    if (sub == 0) {
      s_kind[g] = empty ? 2 : (first ? 1 : 0);
      s_row[g] = (first && !empty) ? hout[(size_t)bin * 32 + g] : 0u;
    }
    __syncthreads();
    if (!empty && !(e & E_SKIP)) {
      int g2 = g;
      while (g2 > 0 && s_kind[g2] == 0) --g2;
      dest[e & E_POS] = prow_dest[s_row[g2]];
    }
    __syncthreads();
  }
}

extern "C" int tfra_plan_partition(const tfra_sparse_plan_t* pl, tfra_workspace_t* ws, int num_shards, int mode,
                                   int64_t* keys_out, int32_t* perm_out, int64_t* d_counts, tfra_stream_t stream) {
  if (!pl || pl->n == 0) return set_error(TFRA_ERR_INVALID, "plan_partition: no built plan");
  if (pl->kind == 1) return set_error(TFRA_ERR_UNSUPPORTED, "plan_partition: needs a plan built with the table's dim (an assign-only plan keeps no positions)");
  return tfra_partition(ws, pl->n, reinterpret_cast<const int64_t*>(pl->d_counts + 32), (const int64_t*)pl->dkeys, num_shards, mode,
                        keys_out, perm_out, d_counts, stream);
}

extern "C" int tfra_plan_positions_to(const tfra_sparse_plan_t* pl, const int32_t* perm, int32_t* dest_out, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!pl || pl->n == 0) return set_error(TFRA_ERR_INVALID, "plan_positions_to: no built plan");
  if (pl->kind == 1) return set_error(TFRA_ERR_UNSUPPORTED, "plan_positions_to: needs a plan built with the table's dim (an assign-only plan keeps no positions)");
  if (!perm || !dest_out) return set_error(TFRA_ERR_INVALID, "plan_positions_to: null buffer");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != pl->device) { if (hipSetDevice(pl->device) != hipSuccess) return set_error(TFRA_ERR_HIP, "plan_positions_to: hipSetDevice"); } }
  unsigned key_blocks, bin_blocks;
  plan_grids(pl, &key_blocks, &bin_blocks);
  plan_dest_keys_kernel<<<(unsigned)std::min<size_t>(2048, (pl->n + 255) / 256), 256, 0, s>>>(keys_of(pl), perm, dest_out, pl->prow_dest);
  plan_dest_bins_kernel<<<bin_blocks, NTA, 0, s>>>(pl->out.hent, pl->out.hout, pl->binmap, pl->d_counts, pl->prow_dest, dest_out);
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "plan_positions_to: launch failed");
  return TFRA_OK;
}

namespace tfra {
void destroy_workspace_plan(void* plan) { if (plan) tfra_sparse_plan_destroy(reinterpret_cast<tfra_sparse_plan*>(plan)); }
}  // namespace tfra

// ---------------------------------------------------------------------------------------------
// One training step driven from C on two streams (no Python between the launches, no graph).
//   main : lookup(ids_cur) -> write-back of batch cur (hot sums + fused update, or assign)      (plan_cur)
//   side : build plan_next from ids_next, free-running
// Cross-queue events cost ~5 us (stream wait) / ~7 us (record) each between two kernels of the main stream,
// so the two streams are ordered through two host-visible counters in pinned memory instead, and the host
// only falls back to a sync when a counter lags:
//   * table progress: written by the first block of the write-back of step s  =>  every earlier step is done.
//     plan_next's buffers were last read by step plan_next->last_used_step; the build is enqueued once the
//     progress has passed it (with >= 3 plans in rotation that is always the case unless the host is far
//     ahead of the GPU, in which case it waits here instead of in a queue);
//   * plan built: generation + counts written by the build's last kernel.  If they already show plan_cur's
//     generation the write-back is enqueued without any wait packet and with exact grids; otherwise — the host
//     got ahead of the side stream — the host waits for the side stream.
static int step_prefetch_impl(tfra_table_t* tp, const tfra_opt_params* p, tfra_sparse_plan_t* plan_cur,
                              const int64_t* ids_cur, void* rows_out, const void* find_default, const void* grads_or_values,
                              const float* param_default_row, const uint64_t* scores, tfra_sparse_plan_t* plan_next,
                              const int64_t* ids_next, size_t n_next, tfra_stream_t main_stream, tfra_stream_t side_stream) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !plan_cur) return set_error(TFRA_ERR_INVALID, "step_prefetch: null argument");
  hipStream_t ms = (hipStream_t)main_stream, ss = (hipStream_t)side_stream;
  if (ms == ss && plan_next) return set_error(TFRA_ERR_INVALID, "step_prefetch: needs two different streams");
  if (plan_next == plan_cur) return set_error(TFRA_ERR_INVALID, "step_prefetch: plan_next must differ from plan_cur");
  std::lock_guard<std::mutex> step_lock(t->step_mu);   // one driver call at a time per table
  int rc = TFRA_OK;
  if (!t->progress_host) {
    if (hipHostMalloc((void**)&t->progress_host, 64, hipHostMallocDefault) != hipSuccess) { t->progress_host = nullptr; return set_error(TFRA_ERR_OOM, "step_prefetch: hipHostMalloc"); }
    t->progress_host[0] = 0; t->progress_host[1] = 0;
  }
  const unsigned step = ++t->step_gen;
  if (plan_next) {
    if (!plan_next->host_counts) {
      if (hipHostMalloc((void**)&plan_next->host_counts, 64, hipHostMallocDefault) != hipSuccess) { plan_next->host_counts = nullptr; return set_error(TFRA_ERR_OOM, "step_prefetch: hipHostMalloc"); }
      for (int i = 0; i < 8; ++i) plan_next->host_counts[i] = 0;
    }
    if (plan_next->last_used_step) {  // the write-back that read plan_next's buffers must be over
      const unsigned need = plan_next->last_used_step + 1;
      volatile unsigned* prog = t->progress_host;
      bool ok = false;
      for (int it = 0; it < 200000 && !ok; ++it) ok = (int)(*prog - need) >= 0;   // ~ a few ms at most
      if (!ok && hipStreamSynchronize(ms) != hipSuccess) return set_error(TFRA_ERR_HIP, "step_prefetch: sync");
    }
  }
  if (plan_cur->n && rows_out) {
    rc = tfra_table_find(tp, plan_cur->n, ids_cur, rows_out, nullptr, find_default, 0, main_stream);
    if (rc) return rc;
  }
  if (plan_next) {
    if (!p) plan_next->skip_counts_once = !(t->opts.strategy == TFRA_EVICT_LFU && !scores);   // (the next step's call passes scores or not like this one)
    rc = tfra_sparse_plan_build(plan_next, n_next, ids_next, p ? t->opts.dim : 0, side_stream);
    if (rc) return rc;
    if (!plan_next->built_ev && hipEventCreateWithFlags(&plan_next->built_ev, hipEventDisableTiming) != hipSuccess) {
      plan_next->built_ev = nullptr;
      return set_error(TFRA_ERR_HIP, "step_prefetch: event");
    }
    if (hipEventRecord(plan_next->built_ev, ss) != hipSuccess) return set_error(TFRA_ERR_HIP, "step_prefetch: event record");
    plan_next->ev_recorded = true;   // built on the side stream: the join below applies
  }
  if (plan_cur->ev_recorded) {  // built on the side stream by an earlier call
    // complete = every block of the build has ended and its stores are in memory (the pinned counts alone do not say that)
    if (hipEventQuery(plan_cur->built_ev) != hipSuccess && hipEventSynchronize(plan_cur->built_ev) != hipSuccess)   // the wait: rare
      return set_error(TFRA_ERR_HIP, "step_prefetch: join");
    plan_cur->ev_recorded = false;
  }
  plan_cur->last_used_step = step;
  if (plan_cur->n == 0) return TFRA_OK;   // no kernel publishes this step: a later slot check falls back to a sync
  std::lock_guard<std::mutex> lock(t->mu);
  if (p) return apply_planned_impl(tp, p, plan_cur, (const float*)grads_or_values, param_default_row, main_stream, t->progress_host, step);
  return upsert_planned_impl(tp, plan_cur, grads_or_values, scores, main_stream, t->progress_host, step);
}

extern "C" int tfra_table_step_prefetch(tfra_table_t* tp, const tfra_opt_params* p, tfra_sparse_plan_t* plan_cur,
                                        const int64_t* ids_cur, void* rows_out, const void* find_default,
                                        const float* grads, const float* param_default_row,
                                        tfra_sparse_plan_t* plan_next, const int64_t* ids_next, size_t n_next,
                                        tfra_stream_t main_stream, tfra_stream_t side_stream) {
  if (!p) return set_error(TFRA_ERR_INVALID, "step_prefetch: null optimizer parameters");
  return step_prefetch_impl(tp, p, plan_cur, ids_cur, rows_out, find_default, grads, param_default_row, nullptr, plan_next, ids_next,
                            n_next, main_stream, side_stream);
}

extern "C" int tfra_table_step_prefetch_assign(tfra_table_t* tp, tfra_sparse_plan_t* plan_cur, const int64_t* ids_cur,
                                               void* rows_out, const void* find_default, const void* values,
                                               const uint64_t* scores, tfra_sparse_plan_t* plan_next,
                                               const int64_t* ids_next, size_t n_next, tfra_stream_t main_stream,
                                               tfra_stream_t side_stream) {
  return step_prefetch_impl(tp, nullptr, plan_cur, ids_cur, rows_out, find_default, values, nullptr, scores, plan_next, ids_next,
                            n_next, main_stream, side_stream);
}

// tf.unique WITHOUT the first-occurrence order (which nothing on the embedding path observes: the distinct ids feed Find / Insert,
// the inverse index feeds the gather — PY/dynamic_embedding_ops.py:99-117, PY/shadow_embedding_ops.py:316): the SET plan of the
// ids IS their de-duplication — one launch — and a second launch turns it into the inverse index.  Two launches instead of the
// three (formerly six) of tfra_unique, no look-back chain.
extern "C" int tfra_unique_unordered(tfra_workspace_t* ws, size_t n, const int64_t* ids, int64_t* unique_out, int32_t* idx_out,
                                     int64_t* d_num_unique, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ws || !d_num_unique) return set_error(TFRA_ERR_INVALID, "unique: null argument");
  { int cur_ = -1; if (hipGetDevice(&cur_) != hipSuccess || cur_ != ws->device) { if (hipSetDevice(ws->device) != hipSuccess) return set_error(TFRA_ERR_HIP, "unique: hipSetDevice"); } }
  if (n == 0) { if (hipMemsetAsync(d_num_unique, 0, sizeof(int64_t), s) != hipSuccess) return set_error(TFRA_ERR_HIP, "unique: memset"); return TFRA_OK; }
  if (!ids || !unique_out || !idx_out) return set_error(TFRA_ERR_INVALID, "unique: null buffer");
  if (n > MAX_IDS) return set_error(TFRA_ERR_UNSUPPORTED, "unique_unordered: at most 2^18 ids per call (tfra_unique takes more)");
  if (!ws->h_err && hipHostMalloc(reinterpret_cast<void**>(&ws->h_err), 64, hipHostMallocDefault) == hipSuccess) *ws->h_err = 0;
  if (ws->h_err && __atomic_load_n(ws->h_err, __ATOMIC_RELAXED)) {   // an EARLIER one-launch call gave up waiting for another block's index
    __atomic_store_n(ws->h_err, 0u, __ATOMIC_RELAXED);
    return set_error(TFRA_ERR_HIP, "unique_unordered: an earlier call on this workspace timed out waiting for a block's index (its idx_out holds -1 entries)");
  }
  if (!ws->uplan) {
    tfra_sparse_plan* pl = nullptr;
    int rc = tfra_sparse_plan_create(ws->device, &pl);
    if (rc) return rc;
    ws->uplan = pl;
  }
  tfra_sparse_plan* pl = reinterpret_cast<tfra_sparse_plan*>(ws->uplan);
  SetPlanLaunch L;
  int rc = setplan_prepare(pl, n, s, false, &L);
  if (rc) return rc;
  if (L.blocks <= 128) {   // ONE launch (all blocks co-resident on half the chip: a block may wait for another's index)
    setplan_kernel<false, true, true><<<L.blocks, SP_NT, 0, s>>>(n, (const i64*)ids, L.m2, L.cur, L.old, L.next_use_count, (i64*)unique_out, idx_out,
                                                                 (i64*)d_num_unique, ws->h_err);
  } else {
    setplan_kernel<false, true><<<L.blocks, SP_NT, 0, s>>>(n, (const i64*)ids, L.m2, L.cur, L.old, L.next_use_count);
    unique_idx_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(n, (const i64*)ids, SetProbe{L.cur.ent, L.m2}, L.cur.ukeys, L.cur.count, (i64*)unique_out,
                                                                 idx_out, (i64*)d_num_unique);
  }
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "unique_unordered: launch failed");
  return TFRA_OK;
}

// TFRA>HkvHashTableEmbeddingLookup (tf_ops/fused_ops_rocm.cc) as ONE launch: Find of all n ids (tfra_table_find's results, default fill
// included) side by side with tfra_unique_unordered of the same ids (find_unique_kernel).  Shapes the one-launch de-duplication does
// not take (rows that are not 16-byte granules) go through the two calls one after the other: same results.
extern "C" int tfra_table_find_unique(tfra_table_t* tp, tfra_workspace_t* ws, size_t n, const int64_t* ids, void* rows_out, uint8_t* exists,
                                      const void* defaults, int default_is_full, int64_t* unique_out, int32_t* idx_out,
                                      int64_t* d_num_unique, tfra_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!tp || !ws || !d_num_unique) return set_error(TFRA_ERR_INVALID, "find_unique: null argument");
  Table* t = reinterpret_cast<Table*>(tp);
  if (t->device != ws->device) return set_error(TFRA_ERR_INVALID, "find_unique: table and workspace live on different devices");
  bool one = n != 0 && n <= MAX_IDS && n <= 256u * SP_NT && ids && rows_out && defaults && unique_out && idx_out;
  if (one) {
    const size_t x = (size_t)t->field_bytes | (size_t)(uintptr_t)rows_out | (size_t)(uintptr_t)defaults;
    one = (x & 15) == 0;
  }
  if (!one) {
    int rc = tfra_table_find(tp, n, ids, rows_out, exists, defaults, default_is_full, stream);
    if (rc) return rc;
    return tfra_unique_unordered(ws, n, ids, unique_out, idx_out, d_num_unique, stream);
  }
  std::lock_guard<std::mutex> lock(t->mu);
  { int rc = t->enter(s); if (rc) return rc; }
  if (!ws->uplan) {
    tfra_sparse_plan* pl = nullptr;
    int rc = tfra_sparse_plan_create(ws->device, &pl);
    if (rc) return rc;
    ws->uplan = pl;
  }
  tfra_sparse_plan* pl = reinterpret_cast<tfra_sparse_plan*>(ws->uplan);
  SetPlanLaunch L;
  int rc = setplan_prepare(pl, n, s, false, &L);
  if (rc) return rc;
  const TableView v = t->view_of(t->cur);
  const unsigned fblocks = (unsigned)((n + 16 * (SP_NT / 64) - 1) / (16 * (SP_NT / 64)));   // 16 keys per wave, 16 waves per block
  // one id per thread of a de-duplicating block up to 131072 ids, two above (<= 128 blocks by default; TFRA_FU_IPT=1 above 131072 ids: up to
  // 256, still co-resident).  Two everywhere (TFRA_FU_IPT=2, tuning)
  // makes this launch 0.3 us shorter and the Insert that follows 1.4 us longer (same box, twice): the order of the distinct ids changes
  static const int ipt_env = [] { const char* e = getenv("TFRA_FU_IPT"); return e ? atoi(e) : 0; }();
  const int ipt = ipt_env == 1 || ipt_env == 2 ? ipt_env : (n <= 128u * SP_NT ? 1 : 2);
  const unsigned ublocks = (unsigned)((n + SP_NT * ipt - 1) / (SP_NT * ipt));   // <= 128 (256 with TFRA_FU_IPT=1): dispatched first, co-resident whatever the find's blocks do
  const unsigned grid = ublocks + fblocks;
#define FU_LAUNCH(PF1, IPT) find_unique_kernel<16, PF1, IPT><<<grid, SP_NT, 0, s>>>(ublocks, n, (const i64*)ids, L.m2, L.cur, L.old, L.next_use_count, \
    (i64*)unique_out, idx_out, (i64*)d_num_unique, v, (unsigned char*)rows_out, exists, (const unsigned char*)defaults, default_is_full)
  if (t->dense) { if (ipt == 1) FU_LAUNCH(true, 1); else FU_LAUNCH(true, 2); }
  else { if (ipt == 1) FU_LAUNCH(false, 1); else FU_LAUNCH(false, 2); }
#undef FU_LAUNCH
  if (hipGetLastError() != hipSuccess) return set_error(TFRA_ERR_HIP, "find_unique: launch failed");
  return TFRA_OK;
}

#define TFRA_STEP_HOST_PART
#include "tfra_step_impl.h"
#undef TFRA_STEP_HOST_PART
