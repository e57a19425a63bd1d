/* tfra_mi355x.h — C ABI of the MI355X-native dynamic-embedding table engine.
 *
 * This is the drop-in boundary: exactly the surface TFRA's GPU adapter
 * (`gpu::TableWrapper<K,V>`) needs from its storage engine, plus the fused extras that
 * replace multi-op sequences of the reference.  Plain pointers and sizes only; all device
 * pointers are HIP device pointers on the table's device; `stream` is a `hipStream_t`
 * passed as `void*`.  Every entry point returns 0 on success or a negative tfra_status and
 * never throws; `tfra_last_error()` returns a thread-local message.  All table operations
 * are STREAM-ORDERED and do not synchronise the host unless documented.
 *
 * Reference files (R = /root/reference/tensorflow_recommenders_addons/dynamic_embedding/core):
 *   R/kernels/lookup_impl/lookup_table_op_hkv.h   gpu::TableWrapper  (what calls the engine)
 *   R/kernels/hkv_hashtable_op_gpu.cu.cc          HkvHashTableOfTensorsGpu (TF op kernels)
 *   R/kernels/cuckoo_hashtable_op.cc              CuckooHashTableOfTensors (CPU semantics =
 *                                                 the result oracle)
 *   R/ops/hkv_hashtable_ops.cc                    op names / attrs
 */
#ifndef TFRA_MI355X_H_
#define TFRA_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFRA_ABI_VERSION 1

typedef struct tfra_table tfra_table_t;
typedef void* tfra_stream_t; /* hipStream_t */

typedef enum {
  TFRA_OK = 0,
  TFRA_ERR_INVALID = -1,   /* bad argument / shape / dtype (TF: InvalidArgument)          */
  TFRA_ERR_OOM = -2,       /* allocation failed                                            */
  TFRA_ERR_HIP = -3,       /* HIP runtime error (TF: Internal)                             */
  TFRA_ERR_FULL = -4,      /* table at max_capacity and key could not be placed            */
  TFRA_ERR_IO = -5,        /* KV-file save/load failed                                     */
  TFRA_ERR_UNSUPPORTED = -6
} tfra_status;

/* value dtypes registered for the GPU ops: K=int64 x V in {float, int8, int32, int64, half,
 * bfloat16} (R/kernels/hkv_hashtable_op_gpu.cu.cc:1133-1138); double is the CPU table's extra
 * (R/kernels/cuckoo_hashtable_op.cc:995-1014). */
typedef enum {
  TFRA_F32 = 0, TFRA_F16 = 1, TFRA_BF16 = 2, TFRA_I8 = 3, TFRA_I32 = 4, TFRA_I64 = 5, TFRA_F64 = 6
} tfra_dtype;

/* eviction strategy enum 0..4 as in R/kernels/lookup_impl/lookup_table_op_hkv.h:454-475 and
 * PY/dynamic_embedding_creator.py:141-146; -1 = never evict (libcuckoo semantics: grow). */
typedef enum {
  TFRA_EVICT_NONE = -1, TFRA_EVICT_LRU = 0, TFRA_EVICT_LFU = 1, TFRA_EVICT_EPOCHLRU = 2,
  TFRA_EVICT_EPOCHLFU = 3, TFRA_EVICT_CUSTOMIZED = 4
} tfra_evict_strategy;

/* Mirrors TableWrapperInitOptions (R/kernels/lookup_impl/lookup_table_op_hkv.h:304-315) and
 * the creator-op attrs (R/ops/hkv_hashtable_ops.cc:318-339).  POD, versioned by struct_size. */
typedef struct {
  uint32_t struct_size;           /* = sizeof(tfra_table_opts)                               */
  int32_t value_dtype;            /* tfra_dtype                                              */
  int32_t dim;                    /* value_shape[0]                                          */
  int32_t aux_fields;             /* S co-located per-key state vectors of `dim` elements
                                     (optimizer slots m,v / accum,linear); 0 for a plain table */
  uint64_t init_capacity;         /* attr init_capacity / init_size; 0 -> 1 Mi
                                     (hkv_hashtable_op_gpu.cu.cc:53) or 8192 for cuckoo flavour */
  uint64_t max_capacity;          /* attr max_capacity; 0 = unbounded (libcuckoo flavour)   */
  uint64_t max_hbm_for_vectors;   /* accepted for attr parity; this engine is pure-HBM       */
  float max_load_factor;          /* growth trigger; 0 -> 0.5 like lookup_table_op_hkv.h:449
                                     when max_capacity>0, 0.75 for the unbounded flavour     */
  int32_t strategy;               /* tfra_evict_strategy                                     */
  int64_t step_per_epoch;         /* EPOCH* strategies (lookup_table_op_hkv.h:528-536)       */
  int32_t reserved_key_start_bit; /* accepted for attr parity: this engine stores EVERY int64
                                     key (libcuckoo parity), nothing is reserved             */
  int32_t device;                 /* HIP device ordinal; -1 = current                        */
  float aux_init[4];              /* initial value of aux field f for a newly inserted key   */
} tfra_table_opts;

/* Allocation bridge = TFOrDefaultAllocator (lookup_table_op_hkv.h:329-426): lets the host
 * framework own the bytes.  NULL -> hipMalloc/hipFree. kind: 0 device, 1 pinned host, 2 host. */
typedef struct {
  void* (*alloc)(void* user, int kind, size_t bytes, tfra_stream_t stream);
  void (*free)(void* user, int kind, void* ptr, tfra_stream_t stream);
  void* user;
} tfra_allocator;

/* flags for insert / accum */
#define TFRA_FLAG_UNIQUE_KEYS 1u /* caller guarantees no duplicate keys in this call (HKV's
                                    contract, PY/dynamic_embedding_variable.py:1377-1378):
                                    single-pass fast path. Without it duplicates resolve like the
                                    single-threaded reference: the LAST occurrence wins.        */

const char* tfra_last_error(void);
int tfra_abi_version(void);

/* -- lifetime: HashTableGpuOp / ResourceMgr own one handle per (shard, variable)
 *    (R/kernels/cuckoo_hashtable_op_gpu.h:43-139); init = lookup_table_op_hkv.h:435-520 ------ */
int tfra_table_create(const tfra_table_opts* opts, const tfra_allocator* alloc, tfra_table_t** out);
int tfra_table_destroy(tfra_table_t* t);

/* -- find = TableWrapper::get (lookup_table_op_hkv.h:719-732) with the default pre-fill FUSED:
 *    values[i,:] = hit ? row : (default_is_full ? defaults[i,:] : defaults[0,:]);
 *    exists[i] (may be NULL) = hit.  Duplicate keys allowed.  Never inserts.
 *    Result oracle: TableWrapperOptimized::find (lookup_table_op_cpu.h:188-217).           */
int tfra_table_find(tfra_table_t* t, size_t n, const int64_t* keys, void* values, uint8_t* exists,
                    const void* defaults, int default_is_full, tfra_stream_t stream);

/* -- insert_or_assign = TableWrapper::upsert (lookup_table_op_hkv.h:522-537); scores NULL or
 *    [n] (HkvHashTableInsert's `scores` input, empty tensor -> NULL).  Advances the epoch
 *    counter for EPOCH* strategies.  Oracle: LaunchTensorsInsert (cuckoo_hashtable_op.cc:111).
 *    With TFRA_FLAG_UNIQUE_KEYS (what the Insert op passes: HKV's contract) the call is ONE pass
 *    with bucket ownership over the caller's keys (the kernels of tfra_table_upsert_planned, no
 *    plan); a bulk load — so many keys for the table's size that most would collide on a home
 *    bucket — and calls without the flag take the locked kernels.  Same results either way.     */
int tfra_table_insert_or_assign(tfra_table_t* t, size_t n, const int64_t* keys, const void* values,
                                const uint64_t* scores, uint32_t flags, tfra_stream_t stream);

/* -- The same two ops with the key COUNT on the device: n = the length of the buffers, *d_n (a device int64, or pinned host
 *    memory the device can read) = the number of keys in them; min(n, *d_n) keys are looked up / written.  For the chain
 *    tfra_unique_unordered -> Find -> gather -> Insert of embedding_lookup (python/ops/dynamic_embedding_ops.py:99-117), where the
 *    count is only the SHAPE of tf.unique's output: the host reads it while these calls already run.  insert_or_assign_n takes
 *    unique keys (TFRA_FLAG_UNIQUE_KEYS semantics, the Insert op's contract) through the single-pass write-back and returns
 *    TFRA_ERR_UNSUPPORTED when that pass cannot take the call (owner tags off, a bulk load): read the count and call the plain
 *    entry point then.  A growing table sizes itself by n, the upper bound.                                                    */
int tfra_table_find_n(tfra_table_t* t, size_t n, const int64_t* d_n, const int64_t* keys, void* values, uint8_t* exists,
                      const void* defaults, int default_is_full, tfra_stream_t stream);
int tfra_table_insert_or_assign_n(tfra_table_t* t, size_t n, const int64_t* d_n, const int64_t* keys, const void* values,
                                  const uint64_t* scores, tfra_stream_t stream);

/* -- accum_or_assign = TableWrapper::accum (lookup_table_op_hkv.h:539-546):
 *    absent & !exists -> insert row; present & exists -> row += delta (element order 0..dim-1,
 *    one add each); otherwise no-op.  Oracle: accumrase_fn (lib/cuckoo/cuckoohash_map.hh:619). */
int tfra_table_accum_or_assign(tfra_table_t* t, size_t n, const int64_t* keys,
                               const void* values_or_deltas, const uint8_t* exists,
                               const uint64_t* scores, uint32_t flags, tfra_stream_t stream);

/* -- erase / clear / size / capacity (lookup_table_op_hkv.h:745-756) ---------------------- */
int tfra_table_erase(tfra_table_t* t, size_t n, const int64_t* keys, tfra_stream_t stream);
int tfra_table_clear(tfra_table_t* t, tfra_stream_t stream);
/* host result: synchronises `stream` (HkvHashTableOfTensorsGpu::size, :162-170) */
int tfra_table_size(tfra_table_t* t, size_t* out, tfra_stream_t stream);
/* device scalar result, no host sync (size_i64, :172-179) */
int tfra_table_size_to_device(tfra_table_t* t, int64_t* d_out, tfra_stream_t stream);
/* Keys the device could not place since the last check (table full at max_capacity with nothing evictable, rehash
 * failure, a write-back plan whose internal capacity overflowed): returns TFRA_ERR_FULL with the count in
 * tfra_last_error() and clears the counter; TFRA_OK when there were none.  Synchronises `stream`.  tfra_table_size
 * performs the same check; the stream-ordered entry points never read the counter themselves. */
int tfra_table_check_errors(tfra_table_t* t, tfra_stream_t stream);
/* number of slots export_batch scans (= value to loop `offset` up to) */
int tfra_table_capacity(tfra_table_t* t, size_t* out);
/* Growth so far (introspection): out4 = {growths, of which in place, storage is a mapped address range (0/1), bytes
 * mapped}.  Tables of TFRA_VMM_THRESHOLD_MB (env, default 4096) or more live in a reserved virtual address range and grow
 * by mapping more memory behind the table and splitting every bucket into its children where it is — peak memory = the
 * new size; smaller tables, and tables on a caller-supplied allocator, grow by copying (old + new coexist). */
int tfra_table_growth_stats(tfra_table_t* t, uint64_t* out4);
/* introspection (tests, tools): out5 = {empty key slots, slots locked by an eviction in progress (0 whenever no
 * call is running), live key slots, buckets with the OVF0 flag, buckets with the OVF1 flag}.  Host buffer;
 * synchronises `stream`. */
int tfra_table_slot_census(tfra_table_t* t, uint64_t* out5, tfra_stream_t stream);
/* grow (rehash) so that at least `min_slots` slots exist; no-op if already that large */
int tfra_table_reserve(tfra_table_t* t, size_t min_slots, tfra_stream_t stream);

/* -- export_batch(n, offset, d_counter, keys, values, scores) (lookup_table_op_hkv.h:548-594):
 *    scans slots [offset, offset+n) and appends live (key, row[, score]) triples at
 *    *d_counter (device size_t, caller zeroes it), order unspecified.  values/scores may be
 *    NULL.  Output buffers must hold every live entry of the range.                        */
int tfra_table_export_batch(tfra_table_t* t, size_t n, size_t offset, size_t* d_counter,
                            int64_t* keys, void* values, uint64_t* scores, tfra_stream_t stream);

/* -- run-time options.  TFRA_OPTION_CAPTURE_SAFE = 1 makes every table entry point safe to call
 *    while `stream` is being captured into a hipGraph (no host synchronisation, no event
 *    record/query, no growth): the caller guarantees capacity (tfra_table_reserve beforehand) and
 *    keeps using ONE stream.  Scratch buffers must have been sized by an identical warm-up call.
 *    TFRA_OPTION_NO_OWNER_TAGS = 1 makes the planned write-backs take the general two-kernel path
 *    (the one used when the 4 B/bucket owner-tag array cannot be allocated); same results.      */
typedef enum { TFRA_OPTION_CAPTURE_SAFE = 1, TFRA_OPTION_NO_OWNER_TAGS = 2, TFRA_OPTION_KEY_BYTES_ON_DISK = 3 } tfra_option;
/*    TFRA_OPTION_KEY_BYTES_ON_DISK = 4 | 8 (default 8): the width of a key in `<prefix>-keys` files.  4 = tables whose OP-LEVEL key type
 *    is int32 (the reference's (int32, float) GPU kernels, K/cuckoo_hashtable_op_gpu.cu.cc:1058: its files hold raw int32 keys): save
 *    narrows, load widens (sign-extending); a key outside the int32 range fails the save. */
int tfra_table_set_option(tfra_table_t* t, int option, int64_t value);

/* -- set_global_epoch (lookup_table_op_hkv.h:499,507,533) --------------------------------- */
int tfra_table_set_global_epoch(tfra_table_t* t, uint64_t epoch);

/* -- KV files: <prefix>-keys (raw int64[]) and <prefix>-values (raw V[n*dim]), native endian
 *    (cuckoo_hashtable_op.cc:310-505; lookup_table_op_hkv.h:602-717).  buffer_keys = keys per
 *    I/O chunk.  load does NOT clear (the GPU op clears first, the CPU op does not).       */
int tfra_table_save(tfra_table_t* t, const char* prefix, size_t buffer_keys, int append,
                    tfra_stream_t stream, size_t* n_saved);
int tfra_table_load(tfra_table_t* t, const char* prefix, size_t buffer_keys, tfra_stream_t stream,
                    size_t* n_loaded);

/* ============================ fused extras (beyond HKV's surface) ========================= */

/* Field access for tables created with aux_fields > 0: same as find / insert_or_assign but on
 * state vector `field` (0 = the embedding itself).  Lets `<param>/<opt>/<slot>` variables of
 * create_slots (PY/dynamic_embedding_optimizer.py:870-958) be views of one physical table.   */
int tfra_table_find_field(tfra_table_t* t, int field, size_t n, const int64_t* keys, void* values,
                          uint8_t* exists, const void* defaults, int default_is_full,
                          tfra_stream_t stream);
int tfra_table_insert_field(tfra_table_t* t, int field, size_t n, const int64_t* keys,
                            const void* values, uint32_t flags, tfra_stream_t stream);
/* KV files of one state vector: the reference checkpoints every slot variable of create_slots as its own table
 * (`<param>/<opt>/<slot>` -> `<param>_<opt>_<slot>_mht_<i>of<N>-keys/-values`); here the slot is field `field` of the
 * parameter's rows.  Same format as tfra_table_save / _load (field 0 = the embedding itself).  Load the embedding
 * first: a key that is not resident yet is created with the parameter's default row. */
int tfra_table_save_field(tfra_table_t* t, int field, const char* prefix, size_t buffer_keys, int append,
                          tfra_stream_t stream, size_t* n_saved);
int tfra_table_load_field(tfra_table_t* t, int field, const char* prefix, size_t buffer_keys, tfra_stream_t stream,
                          size_t* n_loaded);

/* Fused sparse optimizer write-back: replaces (1+S) finds + dense apply + (1+S) upserts
 * (PY/dynamic_embedding_optimizer.py:165-204) with one pass over rows laid out [p|slot..].
 * keys [n] UNIQUE (use tfra_segment_sum first), grads [n,dim] fp32.  Missing keys are inserted
 * with p = param_defaults (full [n,dim] or broadcast [dim], like find) and slots = aux_init.
 * Requires aux_fields >= the optimizer's slot count.  value_dtype F32, F16 or BF16 (tfra_table_apply_optimizer; gradients
 * and defaults are float32 in every case: the rule runs in fp32 on the up-cast row and slots and the results are rounded
 * to the storage type once); the planned / sparse forms below: F32.                           */
typedef enum { TFRA_OPT_SGD = 0, TFRA_OPT_ADAM = 1, TFRA_OPT_ADAGRAD = 2, TFRA_OPT_FTRL = 3 } tfra_opt_kind;
typedef struct {
  int32_t kind;     /* tfra_opt_kind */
  float lr;         /* Adam: pass lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the host (global step) */
  float beta1, beta2, eps;          /* Adam; Adagrad: eps<0 -> TF1 rule (no eps)             */
  float l1, l2, lr_power;           /* FTRL                                                  */
  const float* d_lr;                /* optional DEVICE scalar overriding `lr` (read by the kernel):
                                       lets a captured hipGraph follow a learning-rate schedule /
                                       Adam's per-step lr_t without re-capturing                */
} tfra_opt_params;
/* d_n: optional DEVICE int64 scalar; when non-NULL only the first min(n, *d_n) keys are applied,
 * so a caller can chain tfra_unique -> tfra_segment_sum -> apply without reading the unique count
 * on the host (n is then the buffer length). */
int tfra_table_apply_optimizer(tfra_table_t* t, const tfra_opt_params* p, size_t n,
                               const int64_t* keys, const float* grads, const void* param_defaults,
                               int default_is_full, const int64_t* d_n, tfra_stream_t stream);

/* The whole backward half of a training step: ids [n] MAY repeat (a raw Zipf batch); gradients of equal
 * ids are summed in a fixed order (deterministic; ids occurring <= 8 times in the batch: strictly in batch order),
 * then one fused update per unique key as above.  = _resource_apply_sparse_duplicate_indices + the write-back
 * sequence (PY/dynamic_embedding_optimizer.py:165-204).  param_default_row: [dim] fp32, used for unseen keys.
 * Requires float32 values, dim % 4 == 0, dim <= 256.  More than 2^18 (262 144) ids per call take a slower path (chunk-wise
 * sums, then every key once; host reads of the counts, scratch allocated per call) with the same result up to the
 * association of the sums. */
int tfra_table_apply_sparse(tfra_table_t* t, const tfra_opt_params* p, size_t n, const int64_t* ids,
                            const float* grads, const float* param_default_row, tfra_stream_t stream);

/* insert_or_assign of a batch whose keys MAY repeat — the LAST occurrence wins, like the reference's sequential
 * LaunchTensorsInsert (K/cuckoo_hashtable_op.cc:104-140) — de-duplicated on the device, also on a bounded (Hkv)
 * table running at max_capacity (eviction needs one writer per key).  values [n,dim] of the table's dtype;
 * scores [n] or NULL (the last occurrence's score; LFU without scores counts the occurrences).  More than 2^18 ids are
 * written chunk after chunk on the stream. */
int tfra_table_upsert_sparse(tfra_table_t* t, size_t n, const int64_t* ids, const void* values,
                             const uint64_t* scores, tfra_stream_t stream);

/* The two calls above split at the point where gradients / values are needed.  Which ids repeat, in which order
 * their gradients are summed and which unique keys the step writes depends on the ids alone, and the ids of a step
 * are known at lookup time (PY/embedding_weights.py keeps them in the TrainableWrapper) or earlier (input
 * pipeline).  tfra_sparse_plan_build does that id-only half on ANY stream — next to the lookup of the same ids, or
 * while the previous step still runs: the batch as CSR-by-key (unique keys, and per key its batch positions in
 * ascending order; keys with many occurrences pre-split into runs of <= 512) — and tfra_table_apply_planned /
 * tfra_table_upsert_planned do the rest when the gradients / values exist.  Results are bit-identical to the
 * one-call forms.  The caller orders build and use (event / same stream) and keeps `plan` untouched until the
 * work using it has finished; a plan can be rebuilt for the next batch afterwards (it owns its device buffers:
 * ~ (90 + dim/2) B per id + 12 MB).  dim: the table's dim, or 0 for a plan only used by upsert_planned: an
 * assign needs the LAST position and the count of every distinct id, not the list of its positions, and such a
 * plan is built by ONE kernel (LDS de-duplication per 1024 ids + a global hash set with atomic max; ~21 MB).
 * A plan object is not internally locked: one thread builds / consumes it at a time (different plans and
 * different tables are independent).                                                                  */
typedef struct tfra_sparse_plan tfra_sparse_plan_t;
int tfra_sparse_plan_create(int device, tfra_sparse_plan_t** out);
int tfra_sparse_plan_destroy(tfra_sparse_plan_t* plan);
int tfra_sparse_plan_build(tfra_sparse_plan_t* plan, size_t n, const int64_t* ids, int dim, tfra_stream_t stream);
int tfra_table_apply_planned(tfra_table_t* t, const tfra_opt_params* p, const tfra_sparse_plan_t* plan,
                             const float* grads, const float* param_default_row, tfra_stream_t stream);
int tfra_table_upsert_planned(tfra_table_t* t, const tfra_sparse_plan_t* plan, const void* values,
                              const uint64_t* scores, tfra_stream_t stream);
/* Introspection (tests, tools): counts[6] = {keys with > 8 occurrences, other keys, partial sums, 512-entry bins,
 * entries of the other keys, build errors}; when keys != NULL also the CSR itself, keys with > 8 occurrences first:
 * keys[i], cnt[i], and positions[] = the batch positions of key 0, of key 1, ... each ascending (cap = length of
 * keys/cnt, positions holds n).  A plan built with dim 0 keeps no positions list: counts = {0, distinct keys, 0, 0, 0, 0},
 * keys / cnt in no particular order, positions must be NULL.  Host buffers; synchronises `stream`. */
int tfra_sparse_plan_read(const tfra_sparse_plan_t* plan, uint32_t* counts, int64_t* keys, uint32_t* cnt,
                          uint32_t* positions, size_t cap, tfra_stream_t stream);

/* One whole training step of a single table on two streams, driven from C (what a TF executor does
 * between the ops of one session.run, without the host framework in the loop):
 *   main : rows_out = find(ids_cur) [n = plan_cur's id count; skipped when rows_out is NULL]
 *          -> tfra_table_apply_planned(plan_cur, grads)            (tfra_table_step_prefetch)
 *          or tfra_table_upsert_planned(plan_cur, values, scores)  (tfra_table_step_prefetch_assign)
 *   side : tfra_sparse_plan_build(plan_next, ids_next), free-running; plan_next may be NULL (last step).
 * plan_cur must have been built from ids_cur (by an earlier call as its plan_next, or by
 * tfra_sparse_plan_build on main_stream).  ids_next must be complete in memory when the call is made (the side
 * stream does not wait for the main stream).  The streams are ordered inside, normally without any
 * cross-queue event: rotate >= 3 plans (4 recommended) so that plan_next's buffers — last read by the
 * step that used it as plan_cur — are idle when it is rebuilt; with fewer the call waits on the host.  */
int tfra_table_step_prefetch(tfra_table_t* t, const tfra_opt_params* p, tfra_sparse_plan_t* plan_cur,
                             const int64_t* ids_cur, void* rows_out, const void* find_default,
                             const float* grads, const float* param_default_row,
                             tfra_sparse_plan_t* plan_next, const int64_t* ids_next, size_t n_next,
                             tfra_stream_t main_stream, tfra_stream_t side_stream);
int tfra_table_step_prefetch_assign(tfra_table_t* t, tfra_sparse_plan_t* plan_cur, const int64_t* ids_cur,
                                    void* rows_out, const void* find_default, const void* values,
                                    const uint64_t* scores, tfra_sparse_plan_t* plan_next,
                                    const int64_t* ids_next, size_t n_next, tfra_stream_t main_stream,
                                    tfra_stream_t side_stream);

/* Many tables on one GPU (BASELINE configs[4]: 26 embedding tables): the step of every table — the arguments of
 * tfra_table_step_prefetch (opt != NULL: grads_or_values = float gradients) or tfra_table_step_prefetch_assign (opt ==
 * NULL: grads_or_values = rows of the table's dtype) — issued from a pool of `n_workers` host threads (0 = 4), tables
 * dealt out dynamically.  Give the tables of one call disjoint stream pairs (or a few pairs round-robin) so that their
 * launches and kernels overlap; the call returns when every step has been ENQUEUED (like the single-table form it does
 * not wait for the GPU).  The first error of any table is returned.  In TensorFlow this is the inter-op thread pool
 * running the per-table op sequences of one session.run. */
typedef struct {
  uint32_t struct_size;              /* = sizeof(tfra_step_desc) */
  uint32_t reserved;
  tfra_table_t* table;
  const tfra_opt_params* opt;
  tfra_sparse_plan_t* plan_cur;
  const int64_t* ids_cur;
  void* rows_out;
  const void* find_default;
  const void* grads_or_values;
  const float* param_default_row;
  const uint64_t* scores;
  tfra_sparse_plan_t* plan_next;
  const int64_t* ids_next;
  size_t n_next;
  tfra_stream_t main_stream;
  tfra_stream_t side_stream;
} tfra_step_desc;
int tfra_multi_step_prefetch(size_t n_tables, const tfra_step_desc* descs, int n_workers);

/* -- the overlapped step: lookup of batch i+1 in ONE launch with the write-back of batch i ------------------------------
 * Replaces, for a training loop that alternates Find and Insert on one table, the pair of ops
 *   HkvHashTableOfTensorsGpu::Find   (K/hkv_hashtable_op_gpu.cu.cc:182-213  -> lookup_table_op_hkv.h:719-756 get)
 *   HkvHashTableOfTensorsGpu::Insert (K/hkv_hashtable_op_gpu.cu.cc:253-290  -> lookup_table_op_hkv.h:522-537 upsert)
 * with results IDENTICAL to running them one after the other (Insert exclusive, Find shared: lookup i+1 sees update i).
 * The driver defers the write-back of a batch to the NEXT call and runs it in the same kernel as that call's lookup; ids the
 * two batches share are served from `values_prev` (the rows being written), everything else from the table; an entry the
 * write-back would evict although this lookup looks for it is evicted after the lookup and the lookup's output corrected
 * (csrc/tfra_step_impl.h).  Ids may repeat (the last occurrence wins).  ONE launch per step on ONE stream, nothing
 * waits on the host: steps can be enqueued any number ahead (tfra_table_steps_overlap).  NOT for hipGraph capture: a launch
 * carries host-side rotating state (which of two counter sets the previous launch's tail zeroed, the plan rotation, the
 * ownership generation); a replayed launch would reuse counters nobody zeroed.  TFRA_OPTION_CAPTURE_SAFE tables take the
 * sequential path inside the same entry points.
 *
 *   tfra_table_step_overlap(d, n, ids, rows_out, exists_out, defaults, default_is_full, values_prev, scores_prev,
 *                           n_next, ids_next, n_next2, ids_next2, stream)
 *     rows_out[n, dim] / exists_out[n] (optional) = Find(ids) with the default fill of tfra_table_find;
 *     values_prev [n_prev, dim] = the rows to assign to the ids of the PREVIOUS call (NULL on the first call / after a
 *       flush); they must stay unchanged until this call's work has run;
 *     ids_next / n_next and ids_next2 / n_next2 (optional) = the ids of the next call and of the one after it, as an input
 *       pipeline knows them: the de-duplication plan of a batch is built over TWO launches without atomics (its distinct ids
 *       are scattered into per-window segments by the launch two calls ahead, its table is built by the launch one call ahead).
 *       With ids_next only, the plan of the next batch is built by a launch of its own in front of the step (round 3's plan
 *       kernel); with neither, by the next call in front of ITS step.  ids must stay valid until their batch has been written
 *       back (the call after the one that looked them up).  An announced batch is recognised by (address, length) when its
 *       own call comes: from announcement to write-back its buffer must stay UNMODIFIED (a ring of id buffers needs at least
 *       four entries when ids_next2 is used); a batch that was announced but is never stepped is forgotten by the next call.
 *   tfra_table_step_overlap_flush(d, values_prev, scores_prev, stream) writes the last batch back: call it before the table
 *     is used through any other entry point.
 * Tables or calls the overlap does not cover (LFU / EPOCHLFU / CUSTOMIZED scores — a key may be refused —, caller scores, optimizer slots, rows
 * that are not multiples of 16 bytes) run the same sequence one op after the other inside the same entry points. */
typedef struct tfra_step_driver tfra_step_driver_t;
int tfra_step_driver_create(tfra_table_t* t, tfra_step_driver_t** out);
int tfra_step_driver_destroy(tfra_step_driver_t* d);
int tfra_table_step_overlap(tfra_step_driver_t* d, size_t n, const int64_t* ids, void* rows_out, uint8_t* exists_out,
                            const void* defaults, int default_is_full, const void* values_prev, const uint64_t* scores_prev,
                            size_t n_next, const int64_t* ids_next, size_t n_next2, const int64_t* ids_next2, tfra_stream_t stream);
int tfra_table_step_overlap_flush(tfra_step_driver_t* d, const void* values_prev, const uint64_t* scores_prev, tfra_stream_t stream);
/* `count` consecutive steps from ONE host call (the arguments of tfra_table_step_overlap per step). */
typedef struct {
  uint32_t struct_size;              /* = sizeof(tfra_overlap_step) */
  int32_t default_is_full;
  size_t n;
  const int64_t* ids;
  void* rows_out;
  uint8_t* exists_out;
  const void* defaults;
  const void* values_prev;
  const uint64_t* scores_prev;
  size_t n_next;
  const int64_t* ids_next;
  size_t n_next2;
  const int64_t* ids_next2;
} tfra_overlap_step;
int tfra_table_steps_overlap(tfra_step_driver_t* d, size_t count, const tfra_overlap_step* steps, tfra_stream_t stream);
/* measurement: HIP events around the launch of each of the next `steps` overlapped steps; _kernel_times waits for them and
 * returns the average duration of the step launch in microseconds (rest_kernel_us: a second launch no longer exists, ~0) */
int tfra_step_driver_time_kernels(tfra_step_driver_t* d, size_t steps);
int tfra_step_driver_kernel_times(tfra_step_driver_t* d, double* step_kernel_us, double* rest_kernel_us, size_t* steps);
/* tuning builds (TFRA_STEP_VARIANT & 16): per-role time stamps of the last launches, out[64][6][4], see csrc/tfra_step_impl.h */
int tfra_step_driver_timing(tfra_step_driver_t* d, uint64_t* out);
/* steps taken overlapped / one op after the other so far; whether a write-back is pending; device_counts[3] (optional,
 * synchronises the device) = {evictions the pass deferred because the next lookup wanted the victim, victims the remainder
 * pass noted, output rows it rewrote with the default row} */
int tfra_step_driver_stats(const tfra_step_driver_t* d, uint64_t* overlapped, uint64_t* sequential, int* pending,
                           uint32_t* device_counts, uint32_t* why_sequential, uint64_t* plans_built);
/* lookups whose positions had been sorted one launch ahead (the MAP role of csrc/tfra_step_impl.h: the next batch's positions probed
 * in this batch's plan and packed by where their row comes from — needs ids_next): such a lookup reads no plan entry and, for the
 * positions served from the rows being written, no table line */
int tfra_step_driver_lookups_listed(const tfra_step_driver_t* d, uint64_t* out);
/* plans_built[2] (optional): plans of a next batch built inside the step launch (from pairs scattered one launch earlier: ids
 * known TWO batches ahead) / built by a launch of their own in front of the step (ids known one batch ahead only, or not at all) */
/* why_sequential (optional): why the last step that ran one op after the other did — 1 empty batch, 2 a buffer or the row
 * size is not a multiple of 16 bytes, 4 no owner tags, 8 optimizer slots / a strategy that may refuse a key / caller scores, 16 and 32
 * (round 5: the table can still grow / is not yet dense) are no longer reasons unless TFRA_STEP_GROWING=0, 64 the
 * previous batch was empty, 128 TFRA_OPTION_CAPTURE_SAFE */

/* -- front-end helpers (N1/N3 rows of SURVEY.md §8f) ------------------------------------- */

/* Scratch for unique/partition; grows on demand, reusable across calls on one stream. */
typedef struct tfra_workspace tfra_workspace_t;
int tfra_workspace_create(int device, tfra_workspace_t** out);
int tfra_workspace_destroy(tfra_workspace_t* ws);

/* tf.unique: unique_out[0..*d_num_unique) in order of first occurrence, idx_out[i] = position of
 * ids[i] in unique_out (PY/dynamic_embedding_ops.py:99; PY/shadow_embedding_ops.py:316).
 * d_num_unique is a device int64 scalar (no host sync). */
int tfra_unique(tfra_workspace_t* ws, size_t n, const int64_t* ids, int64_t* unique_out,
                int32_t* idx_out, int64_t* d_num_unique, tfra_stream_t stream);
/* The same without tf.unique's first-occurrence ORDER (unique_out in no particular order, idx_out consistent with it): what
 * embedding_lookup needs of tf.unique (PY/dynamic_embedding_ops.py:99-117 — the distinct ids feed Find / Insert, the inverse
 * index the gather; the order is not observable there).  Two launches instead of three; n <= 2^18.  d_num_unique may be
 * device-visible pinned host memory: the count then reaches the host without a copy. */
int tfra_unique_unordered(tfra_workspace_t* ws, size_t n, const int64_t* ids, int64_t* unique_out,
                          int32_t* idx_out, int64_t* d_num_unique, tfra_stream_t stream);

/* Find of all n ids AND tfra_unique_unordered of the same ids in ONE launch — the forward half of embedding_lookup as the fused TF op
 * TFRA>HkvHashTableEmbeddingLookup issues it (tf_ops/fused_ops_rocm.cc; PY/dynamic_embedding_ops.py:99-117 needs the rows of every id,
 * the distinct ids and the inverse index for the backward pass).  rows_out / exists / defaults / default_is_full as tfra_table_find,
 * unique_out / idx_out / d_num_unique as tfra_unique_unordered (n <= 2^18); the same results as the two calls one after the other (which
 * is what runs for rows that are not 16-byte granules). */
int tfra_table_find_unique(tfra_table_t* t, tfra_workspace_t* ws, size_t n, const int64_t* ids, void* rows_out, uint8_t* exists,
                           const void* defaults, int default_is_full, int64_t* unique_out, int32_t* idx_out,
                           int64_t* d_num_unique, tfra_stream_t stream);

/* out[idx[i],:] += in[i,:] in index order per segment is NOT guaranteed; sums are accumulated in
 * fp32 with a fixed tree per segment => deterministic. out is [num_segments, dim], zeroed here. */
int tfra_segment_sum(tfra_workspace_t* ws, size_t n, int dim, const float* in, const int32_t* idx,
                     const int64_t* d_num_segments, size_t max_segments, float* out,
                     tfra_stream_t stream);

/* tf.unique + unsorted_segment_sum in one call, parallel per key and order-fixed (the reduction half of
 * tfra_table_apply_sparse, same summation tree: PY/dynamic_embedding_optimizer.py:177-190): keys_out[0..*d_count) =
 * the distinct ids (unspecified order, may differ between calls), rows_out[i,:] = sum of the rows of `in` whose
 * id is keys_out[i] (bit-reproducible per key; ids occurring <= 8 times: strictly in input order).  keys_out [n],
 * rows_out [n,dim]; fp32, dim % 4 == 0, dim <= 256, n <= 2^18.  *d_count = -1 if an internal capacity overflowed
 * (pathological hash skew).  Unlike tfra_segment_sum the cost does not grow with the multiplicity of the hottest id. */
int tfra_reduce_by_key(tfra_workspace_t* ws, size_t n, const int64_t* ids, int dim, const float* in,
                       int64_t* keys_out, float* rows_out, int64_t* d_count, tfra_stream_t stream);

/* The reduction half of tfra_reduce_by_key for a batch whose plan was built AHEAD (tfra_sparse_plan_build with dim = the
 * rows' dim, any stream): rows_out[dest[p], :] = sum of the rows of `grads` whose id equals ids[p] (same summation tree,
 * bit-reproducible).  dest [n] int32 maps batch positions to output rows and must be equal for all positions of one id —
 * e.g. position -> owner-major index of the multi-GPU gradient route, which depends on the ids alone and is computed
 * one batch ahead together with the plan (PY/shadow_embedding_ops.py:397-447 does unique + partition at lookup time). */
int tfra_plan_reduce_to(const tfra_sparse_plan_t* plan, const float* grads, const int32_t* dest, float* rows_out,
                        tfra_stream_t stream);

/* The distinct ids of a built plan in place of tfra_unique, for a caller that needs them grouped by owner (the multi-GPU
 * route; PY/shadow_embedding_ops.py:316,397-422 does unique then dynamic_partition):
 *   tfra_plan_partition   = tfra_partition over the plan's distinct keys (order unspecified): keys_out owner-major,
 *                           perm_out[j] = the plan's index of the key in owner-major row j, d_counts[num_shards];
 *   tfra_plan_positions_to: dest_out[p] = j for every batch position p whose id is the key of owner-major row j —
 *                           the position -> row map for tfra_gather_rows (lookup) and tfra_plan_reduce_to (gradients). */
int tfra_plan_partition(const tfra_sparse_plan_t* plan, tfra_workspace_t* ws, int num_shards, int mode,
                        int64_t* keys_out, int32_t* perm_out, int64_t* d_counts, tfra_stream_t stream);
int tfra_plan_positions_to(const tfra_sparse_plan_t* plan, const int32_t* perm, int32_t* dest_out, tfra_stream_t stream);

/* int32 <-> int64 keys on the device (sign-extending / truncating): the engine's keys are int64; a caller whose op-level key type is
 * int32 (K/cuckoo_hashtable_op_gpu.cu.cc:1058 REGISTER_KERNEL(int32, float)) widens in front of every table call and narrows the keys
 * an export returns.  tfra_keys_narrow_i32 counts the keys that do not fit into *d_overflow (device int64, may be NULL). */
int tfra_keys_widen_i32(size_t n, const int32_t* keys_in, int64_t* keys_out, tfra_stream_t stream);
int tfra_keys_narrow_i32(size_t n, const int64_t* keys_in, int32_t* keys_out, int64_t* d_overflow, tfra_stream_t stream);

/* out[i,:] = rows[idx[i],:] (tf.gather after unique). row_bytes = dim*sizeof(V). */
int tfra_gather_rows(size_t n, size_t row_bytes, const void* rows, const int32_t* idx, void* out,
                     tfra_stream_t stream);

/* embedding_lookup_sparse combiner (PY/dynamic_embedding_ops.py:224-291; SparseSegmentSum/Mean/SqrtN,
 * K/segment_reduction_ops_gpu.cu.cc:30-60): out[r,:] = combine_{i : seg[i]==r} w[i] * rows[idx[i],:]
 * with seg ASCENDING (SparseTensor row ids), weights NULL => 1.  combiner 0 sum, 1 mean (/ sum w),
 * 2 sqrtn (/ sqrt(sum w^2)); empty rows -> 0.  rows/out fp32; members are added in input order. */
int tfra_sparse_segment_combine(tfra_workspace_t* ws, size_t nnz, int dim, const float* rows,
                                const int32_t* idx, const int64_t* seg, const float* weights, int combiner,
                                size_t n_rows, float* out, tfra_stream_t stream);

/* default_partition_fn (PY/dynamic_embedding_variable.py:165-197) + dynamic_partition in one
 * pass: owner[i] = mode 0: (key & 0x7fffffff) % num_shards (CUDA-build branch)
 *                  mode 1: floor_mod(key, num_shards)       (CPU-build branch)
 *                  mode 2: fmix64(key) % num_shards         (opt-in, Zipf-balanced)
 * Produces owner-major keys_out, perm_out (original index of each output element, i.e. the
 * dynamic_partition of range(n)) and d_counts[num_shards] (device int64).
 * d_n: optional DEVICE int64 scalar; when non-NULL only the first min(n, *d_n) keys are partitioned
 * (chains after tfra_unique without reading the unique count on the host).                  */
int tfra_partition(tfra_workspace_t* ws, size_t n, const int64_t* d_n, const int64_t* keys, int num_shards,
                   int mode, int64_t* keys_out, int32_t* perm_out, int64_t* d_counts, tfra_stream_t stream);

/* Same, for a caller-computed owner[i] in [0,num_shards) (custom `partitioner=` functions,
 * PY/dynamic_embedding_variable.py:484-500): only perm_out and d_counts are produced. */
int tfra_partition_by_owner(tfra_workspace_t* ws, size_t n, const int32_t* owner, int num_shards,
                            int32_t* perm_out, int64_t* d_counts, tfra_stream_t stream);

/* dynamic_stitch for one flat permutation: out[perm[i],:] = in[i,:]. */
int tfra_scatter_rows(size_t n, size_t row_bytes, const void* in, const int32_t* perm, void* out,
                      tfra_stream_t stream);

/* Size restriction (PY/restrict_policies.py:205-230,332-358 do export -> top_k(-status)
 * -> gather -> remove in Python): keys_out[0..k) = the k keys with the LOWEST status (timestamp or
 * frequency), ties resolved in input order (stable).  status_dtype TFRA_I32 or TFRA_I64; k <= n.
 * Feed keys_out to tfra_table_erase of the parameter table and of the status table.            */
int tfra_select_lowest(tfra_workspace_t* ws, size_t n, const int64_t* keys, const void* status,
                       int status_dtype, size_t k, int64_t* keys_out, tfra_stream_t stream);

/* -- multi-GPU routed step issued from C (the route of HvdAllToAllEmbedding,
 *    PY/shadow_embedding_ops.py:397-447 / DE/python/keras/layers/embedding.py:545-594, prepared ahead of the step).
 *
 *    A transport moves peer-major device buffers between the ranks of one job: send_bytes[r] bytes go to rank r,
 *    recv_bytes[r] bytes arrive from it (an alltoallv), enqueued on `stream`.  channel 0 carries rows and gradients
 *    (the critical path of a step), channel 1 the counts and ids of later batches; the two never wait for each other.
 *    Every rank issues the same sequence of calls.  tfra_rccl_transport_* is the RCCL one (grouped ncclSend/ncclRecv
 *    over xGMI, one communicator per channel; librccl is dlopen()ed from `librccl_path`, the copy the host framework
 *    already loaded); tests plug a host-staged transport into the same driver.                                       */
typedef struct {
  void* ctx;
  int rank, world;
  int (*alltoallv)(void* ctx, int channel, const void* send, const size_t* send_bytes, void* recv,
                   const size_t* recv_bytes, tfra_stream_t stream);
  /* optional (NULL: the driver issues two alltoallv): TWO independent alltoallv as ONE exchange — the RCCL transport groups all their
   * sends / recvs into one ncclGroup, one kernel instead of two (the count exchange of one batch with the id exchange of another) */
  int (*alltoallv2)(void* ctx, int channel, const void* send_a, const size_t* send_bytes_a, void* recv_a, const size_t* recv_bytes_a,
                    const void* send_b, const size_t* send_bytes_b, void* recv_b, const size_t* recv_bytes_b, tfra_stream_t stream);
} tfra_transport;
#define TFRA_RCCL_ID_BYTES 128
int tfra_rccl_unique_id(const char* librccl_path, void* id_out /* TFRA_RCCL_ID_BYTES, host */);
/* ids: 2 x TFRA_RCCL_ID_BYTES (one per channel), made by rank 0 and distributed by the caller; call on every rank */
int tfra_rccl_transport_create(const char* librccl_path, const void* ids, int rank, int world, int device,
                               tfra_transport* out);
int tfra_rccl_transport_destroy(tfra_transport* tr);
/* ranks of the transport's communicators as RCCL itself reports them (ncclCommCount, both channels) */
int tfra_rccl_transport_ranks(const tfra_transport* tr, int* out);

/*    The driver.  feed() hands a batch to the id-only half of the route, which runs ahead of the step on the driver's
 *    own streams: the de-duplication plan of the batch and its distinct ids grouped by owner (a helper thread launches
 *    these), the count exchange, the split sizes on their way to pinned memory, the id exchange, the position ->
 *    returned-row map and the plan of the ids this rank serves.  A batch moves one stage per step, so keep THREE batches
 *    fed ahead of the one being looked up and nothing ever waits (fewer works too: the missing stages then run, and
 *    stall, inside lookup).  Every collective is issued by the calling thread at points that depend on the call
 *    sequence alone, so all ranks issue them in the same order.
 *      lookup : rows_out[i,:] = row of ids[i] of the OLDEST fed batch (owner's find, default row for missing keys,
 *               alltoall of the rows, one gather)
 *      apply  : gradient rows of that batch -> per-key sums in owner-major order -> alltoall -> fused sparse update at
 *               the owner (tfra_table_apply_planned: sums the <= world parts of a key in a fixed order); retires it
 *    transport NULL: one rank, buffers are copied on the device.  fp32 rows, dim % 4 == 0, dim <= 256, batches of at
 *    most max_batch <= 2^18 ids; the ids a rank serves per batch must not exceed 2^18.  One caller thread.
 *    flags: TFRA_ROUTE_NO_THREAD = everything is issued by the calling thread.                                      */
#define TFRA_ROUTE_NO_THREAD 1u
typedef struct tfra_route tfra_route_t;
int tfra_route_create(tfra_table_t* table, const tfra_transport* transport, int partition_mode, size_t max_batch,
                      uint32_t flags, tfra_route_t** out);
int tfra_route_destroy(tfra_route_t* r);
int tfra_route_feed(tfra_route_t* r, size_t n, const int64_t* d_ids, int ids_ready, tfra_stream_t stream);
int tfra_route_lookup(tfra_route_t* r, float* d_rows_out, const float* default_row, tfra_stream_t stream);
int tfra_route_apply(tfra_route_t* r, const tfra_opt_params* p, const float* d_grads, const float* param_default_row,
                     tfra_stream_t stream);
/* ids this rank serves for the oldest fed batch (device pointer valid until its apply; waits for the split sizes) */
int tfra_route_served_ids(tfra_route_t* r, const int64_t** d_ids, size_t* n, size_t* n_distinct_local);

/* -- the metric's step on a hash-sharded table: lookup(B) + insert_or_assign(B) with ids / rows / values routed ---------------
 *    (csrc/tfra_aroute.hip).  Replaces, for a table sharded over the GPUs of one node, the reference's sequence
 *      lookup : HvdAllToAllEmbedding.__alltoall_embedding_lookup__  (PY/shadow_embedding_ops.py:397-447: unique -> partition ->
 *               hvd.alltoall(ids) -> local Find -> hvd.alltoall(rows) -> stitch)
 *      insert : Variable.upsert on a sharded Variable            (PY/dynamic_embedding_variable.py:772-800: keys AND values
 *               partitioned by owner, one Insert per shard; the last occurrence of a repeated key wins)
 *    with TWO launches per batch (route plan: insert + emit) for everything that depends on the ids alone (de-duplication, grouping by owner, last positions,
 *    position -> row map: runs ahead, on the driver's own streams, followed by the count exchange, the one host read of the split
 *    sizes and the id exchange) and, per step on the caller's stream: gather(values at the last positions) -> alltoall(values) ->
 *    tfra_table_step_overlap at the owner (lookup of the ids it serves for THIS batch + write-back of the rows it received for the
 *    PREVIOUS one, one launch) -> alltoall(rows) -> gather(rows -> positions).
 *    Results = ONE table that sees, per step, lookup(every rank's ids) and then insert_or_assign(rank 0's batch, rank 1's, ...):
 *    a key written by several ranks in one step keeps the highest rank's last occurrence; lookup i+1 sees every write of step i.
 *      feed  : announce a batch (ids must stay valid and unchanged until the batch has been written back).  A batch moves one
 *              stage of its route per step: keep FIVE batches fed ahead of the one being looked up and nothing waits on the
 *              host (fewer works too: the missing stages then run, and stall, inside step; the owner's overlapped step then
 *              builds its de-duplication plans by launches of their own).  At most six.
 *      step  : rows_out[i,:] = row of ids[i] of the OLDEST fed batch not yet looked up (default_row for missing keys: one row);
 *              values_prev [n_prev, dim] = the rows to assign to the ids of the batch the PREVIOUS step looked up (NULL on the
 *              first step / after a flush); they must stay unchanged until this call's work has run.
 *      flush : writes the last looked-up batch back.  Call it before the table is used through any other entry point.
 *    transport NULL: one rank (device copies where the alltoalls would be).  Any value type; batches of at most max_batch <= 2^18
 *    ids, at most 2^18 ids served per rank and batch, at most 64 ranks.  Every rank issues the same call sequence; one caller thread.
 *    stats out6: steps, host stalls on split sizes, owner steps taken overlapped / one op after the other, distinct ids of the last
 *    batch looked up, ids this rank served for it. */
typedef struct tfra_assign_route tfra_assign_route_t;
int tfra_assign_route_create(tfra_table_t* table, const tfra_transport* transport, int partition_mode, size_t max_batch,
                             tfra_assign_route_t** out);
int tfra_assign_route_destroy(tfra_assign_route_t* r);
int tfra_assign_route_feed(tfra_assign_route_t* r, size_t n, const int64_t* d_ids, int ids_ready, tfra_stream_t stream);
int tfra_assign_route_step(tfra_assign_route_t* r, void* d_rows_out, const void* default_row, const void* values_prev,
                           tfra_stream_t stream);
int tfra_assign_route_flush(tfra_assign_route_t* r, const void* values_prev, tfra_stream_t stream);
int tfra_assign_route_stats(const tfra_assign_route_t* r, uint64_t* out6);
/* measurement: tfra_step_driver_time_kernels / _kernel_times of the owner's step driver (the step launch of the next `steps` steps) */
int tfra_assign_route_time_kernels(tfra_assign_route_t* r, size_t steps);
int tfra_assign_route_kernel_times(tfra_assign_route_t* r, double* step_kernel_us, size_t* steps);

#ifdef __cplusplus
}
#endif
#endif /* TFRA_MI355X_H_ */
