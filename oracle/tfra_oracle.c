/* TEST INFRASTRUCTURE ONLY — never linked into, imported by or executed from the product
 * path (recommenders-addons_amd/).  Allowed callers: tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg.
 *
 * oracle/libtfra_oracle.so — a plain-C restatement of the RESULTS the reference's CPU
 * dynamic-embedding table produces for the hot path.  It restates semantics, not the
 * cuckoo placement: the reference's tests only ever compare sorted exports
 * (T/dynamic_embedding_variable_test.py:459-466), so bucket layout is free.
 *
 * Pinned against (tests/test_oracle.py):
 *   (1) the reference's own inline KATs K1..K15, and
 *   (2) oracle/_ref/libtfra_ref.so = the reference's real cuckoohash_map.hh compiled in
 *       place (oracle/ref_shim.cc), on seeded random op sequences -> tests/golden/.
 *
 * R = /root/reference/tensorflow_recommenders_addons/dynamic_embedding/core
 * Each function cites the reference lines it follows.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef long long i64;
typedef unsigned long long u64;

enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2, DT_I8 = 3, DT_I32 = 4, DT_I64 = 5, DT_F64 = 6 };

static size_t dt_size(int dt) {
  switch (dt) {
    case DT_F32: case DT_I32: return 4;
    case DT_F16: case DT_BF16: return 2;
    case DT_I8: return 1;
    default: return 8;
  }
}

/* ---- storage: open addressing, linear probing, tombstones; grows at 70 % ---------- */
enum { ST_EMPTY = 0, ST_FULL = 1, ST_DEAD = 2 };
typedef struct {
  int dtype;
  i64 dim;
  size_t row_bytes;
  u64 cap, live, used; /* cap is a power of two; used = live + tombstones */
  i64* keys;
  unsigned char* st;
  unsigned char* rows;
} otable;

/* R/kernels/lookup_impl/lookup_table_op_cpu.h:90-101 HybridHash<int64> (murmur3 fmix64).
 * Only used for placement here. */
static u64 fmix64(u64 k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

static void ot_alloc(otable* t, u64 cap) {
  t->cap = cap; t->live = 0; t->used = 0;
  t->keys = (i64*)malloc(cap * sizeof(i64));
  t->st = (unsigned char*)calloc(cap, 1);
  t->rows = (unsigned char*)malloc(cap * t->row_bytes);
}

static i64 ot_lookup(const otable* t, i64 key) {
  u64 m = t->cap - 1, i = fmix64((u64)key) & m;
  for (;;) {
    if (t->st[i] == ST_EMPTY) return -1;
    if (t->st[i] == ST_FULL && t->keys[i] == key) return (i64)i;
    i = (i + 1) & m;
  }
}

static void ot_grow(otable* t);

/* returns slot; *is_new set when the key was absent */
static i64 ot_claim(otable* t, i64 key, int* is_new) {
  i64 f = ot_lookup(t, key);
  if (f >= 0) { *is_new = 0; return f; }
  if ((t->used + 1) * 10 > t->cap * 7) ot_grow(t);
  u64 m = t->cap - 1, i = fmix64((u64)key) & m;
  while (t->st[i] == ST_FULL) i = (i + 1) & m;
  if (t->st[i] == ST_EMPTY) t->used++;
  t->st[i] = ST_FULL; t->keys[i] = key; t->live++;
  *is_new = 1;
  return (i64)i;
}

static void ot_grow(otable* t) {
  otable o = *t;
  ot_alloc(t, o.live * 4 > o.cap ? o.cap * 2 : o.cap);
  for (u64 i = 0; i < o.cap; ++i)
    if (o.st[i] == ST_FULL) {
      int n; i64 s = ot_claim(t, o.keys[i], &n);
      memcpy(t->rows + (u64)s * t->row_bytes, o.rows + i * o.row_bytes, t->row_bytes);
    }
  free(o.keys); free(o.st); free(o.rows);
}

/* ---- half / bfloat16 <-> float (round-to-nearest-even), for typed accum ----------- */
static float h2f(uint16_t h) {
  uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
  if (e == 0) {
    if (m == 0) u = s;
    else { e = 127 - 15 + 1; while (!(m & 0x400)) { m <<= 1; e--; } u = s | (e << 23) | ((m & 0x3ff) << 13); }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}
static uint16_t f2h(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  uint32_t s = (u >> 16) & 0x8000; int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15; uint32_t m = u & 0x7fffff;
  if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(s | 0x7c00 | (m ? 0x200 : 0));
  if (e >= 31) return (uint16_t)(s | 0x7c00);
  if (e <= 0) {
    if (e < -10) return (uint16_t)s;
    m |= 0x800000; uint32_t shift = (uint32_t)(14 - e);
    uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) r++;
    return (uint16_t)(s | r);
  }
  uint32_t r = (uint32_t)(e << 10) | (m >> 13), rem = m & 0x1fff;
  if (rem > 0x1000 || (rem == 0x1000 && (r & 1))) r++;
  return (uint16_t)(s | r);
}
static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2b(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffff) > 0x7f800000) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

/* R/kernels/lookup_impl/lookup_table_op_cpu.h:42-52 ValueArray::operator+= :
 * element j = 0..dim-1 in order, ONE add per element (so fp32 results are bit-reproducible). */
static void row_add(int dt, i64 dim, void* row, const void* delta) {
  for (i64 j = 0; j < dim; ++j) switch (dt) {
    case DT_F32: ((float*)row)[j] += ((const float*)delta)[j]; break;
    case DT_F64: ((double*)row)[j] += ((const double*)delta)[j]; break;
    case DT_I8:  ((int8_t*)row)[j] = (int8_t)(((int8_t*)row)[j] + ((const int8_t*)delta)[j]); break;
    case DT_I32: ((int32_t*)row)[j] = (int32_t)((uint32_t)((int32_t*)row)[j] + (uint32_t)((const int32_t*)delta)[j]); break;
    case DT_I64: ((i64*)row)[j] = (i64)((u64)((i64*)row)[j] + (u64)((const i64*)delta)[j]); break;
    case DT_F16: ((uint16_t*)row)[j] = f2h(h2f(((uint16_t*)row)[j]) + h2f(((const uint16_t*)delta)[j])); break;
    case DT_BF16: ((uint16_t*)row)[j] = f2b(b2f(((uint16_t*)row)[j]) + b2f(((const uint16_t*)delta)[j])); break;
  }
}

/* ---- C entry points (same shapes as oracle/ref_shim.cc so one binding serves both) - */

/* R/kernels/cuckoo_hashtable_op.cc:187-209 (init_size 0 -> env/8192; only a sizing hint) */
void* tfra_oracle_create(int dtype, i64 dim, u64 init_size) {
  otable* t = (otable*)calloc(1, sizeof(otable));
  t->dtype = dtype; t->dim = dim; t->row_bytes = (size_t)dim * dt_size(dtype);
  if (init_size == 0) init_size = 8192;
  u64 cap = 16; while (cap * 7 < init_size * 10) cap <<= 1;
  ot_alloc(t, cap);
  return t;
}
void tfra_oracle_destroy(void* p) { otable* t = (otable*)p; free(t->keys); free(t->st); free(t->rows); free(t); }

/* LaunchTensorsFind{,WithExists} (R/kernels/cuckoo_hashtable_op.cc:39-106) over
 * TableWrapperOptimized::find (R/kernels/lookup_impl/lookup_table_op_cpu.h:188-217):
 * hit -> copy row; miss -> is_full_default ? default[i,:] : default[0,:].
 * `is_full_default` is decided by the CALLER from element counts (:48-50). `threads` is
 * accepted for signature parity with the ref shim and ignored (results do not depend on it). */
void tfra_oracle_find(void* p, i64 n, const i64* keys, void* values, const void* defaults,
                      int is_full_default, unsigned char* exists, int threads) {
  otable* t = (otable*)p; (void)threads;
  for (i64 i = 0; i < n; ++i) {
    i64 s = ot_lookup(t, keys[i]);
    unsigned char* out = (unsigned char*)values + (size_t)i * t->row_bytes;
    if (s >= 0) memcpy(out, t->rows + (u64)s * t->row_bytes, t->row_bytes);
    else memcpy(out, (const unsigned char*)defaults + (is_full_default ? (size_t)i * t->row_bytes : 0), t->row_bytes);
    if (exists) exists[i] = s >= 0;
  }
}

/* LaunchTensorsInsert (R/kernels/cuckoo_hashtable_op.cc:111-150) -> insert_or_assign
 * (R/lib/cuckoo/cuckoohash_map.hh:735-740): sequential, so the LAST duplicate wins.
 * clear!=0 is ImportValues = clear + insert (:288-291). */
void tfra_oracle_insert(void* p, i64 n, const i64* keys, const void* values, int clear, int threads) {
  otable* t = (otable*)p; (void)threads;
  if (clear) { memset(t->st, 0, t->cap); t->live = t->used = 0; }
  for (i64 i = 0; i < n; ++i) {
    int is_new; i64 s = ot_claim(t, keys[i], &is_new);
    memcpy(t->rows + (u64)s * t->row_bytes, (const unsigned char*)values + (size_t)i * t->row_bytes, t->row_bytes);
  }
}

/* LaunchTensorsAccum (R/kernels/cuckoo_hashtable_op.cc:155-182) -> insert_or_accum ->
 * accumrase_fn (R/lib/cuckoo/cuckoohash_map.hh:619-633,755-765):
 *   absent  & !exist -> insert the row          present &  exist -> row += delta
 *   absent  &  exist -> nothing                 present & !exist -> nothing        */
void tfra_oracle_accum(void* p, i64 n, const i64* keys, const void* vod, const unsigned char* exists, int threads) {
  otable* t = (otable*)p; (void)threads;
  for (i64 i = 0; i < n; ++i) {
    const unsigned char* src = (const unsigned char*)vod + (size_t)i * t->row_bytes;
    i64 s = ot_lookup(t, keys[i]);
    if (s < 0 && !exists[i]) {
      int is_new; s = ot_claim(t, keys[i], &is_new);
      memcpy(t->rows + (u64)s * t->row_bytes, src, t->row_bytes);
    } else if (s >= 0 && exists[i]) {
      row_add(t->dtype, t->dim, t->rows + (u64)s * t->row_bytes, src);
    }
  }
}

/* Remove (R/kernels/cuckoo_hashtable_op.cc:268-276): serial; absent keys ignored. */
void tfra_oracle_remove(void* p, i64 n, const i64* keys) {
  otable* t = (otable*)p;
  for (i64 i = 0; i < n; ++i) { i64 s = ot_lookup(t, keys[i]); if (s >= 0) { t->st[s] = ST_DEAD; t->live--; } }
}

/* Clear (R/kernels/cuckoo_hashtable_op.cc:278-281) */
void tfra_oracle_clear(void* p) { otable* t = (otable*)p; memset(t->st, 0, t->cap); t->live = t->used = 0; }

/* size (R/kernels/cuckoo_hashtable_op.cc:213) */
u64 tfra_oracle_size(void* p) { return ((otable*)p)->live; }

/* ExportValues / dump(offset, length) (R/kernels/cuckoo_hashtable_op.cc:293-308;
 * R/kernels/lookup_impl/lookup_table_op_cpu.h:219-252): walk live entries in engine order,
 * skip `offset`, emit up to `length` (all the rest when offset+length >= size); returns 0 when
 * offset > size.  Order is engine-internal in the reference; tests sort. */
u64 tfra_oracle_dump(void* p, i64* keys, void* values, u64 offset, u64 length) {
  otable* t = (otable*)p;
  if (offset > t->live || t->live == 0) return 0;
  u64 seen = 0, out = 0, end = (offset + length >= t->live) ? t->live : offset + length;
  for (u64 i = 0; i < t->cap && seen < end; ++i) {
    if (t->st[i] != ST_FULL) continue;
    if (seen >= offset) {
      keys[out] = t->keys[i];
      memcpy((unsigned char*)values + out * t->row_bytes, t->rows + i * t->row_bytes, t->row_bytes);
      out++;
    }
    seen++;
  }
  return out;
}
