// Host-side table object behind the opaque tfra_table_t handle.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/tfra_mi355x.h"
#include "tfra_device.h"

namespace tfra {

extern thread_local std::string g_last_error;
int set_error(int code, const std::string& msg);

struct Storage {
  unsigned char* base = nullptr;  // nb bucket blocks [key line | score line | 15 rows] + 2 side rows
  u64 nb = 0;
  // big tables: physical memory mapped chunk by chunk into ONE reserved virtual range (hipMemAddressReserve / hipMemMap),
  // so that growth maps more memory behind the table and splits the buckets in place (Table::grow_in_place)
  bool vmm = false;
  size_t va_bytes = 0, mapped = 0, chunk_bytes = 0;   // every chunk has the same size (see vmm_map_more)
  std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> chunks;
};

struct AuxInitPod {
  unsigned elem_bytes;
  unsigned pattern[4];
};

struct Table {
  tfra_table_opts opts{};
  tfra_allocator alloc{nullptr, nullptr, nullptr};
  int device = 0;
  unsigned field_bytes = 0, row_stride = 0;
  Storage cur;
  u64* size_shards = nullptr;
  unsigned* reserved_present = nullptr;  // [2] + err_count + d_scalar in one 64-B block
  unsigned* err_count = nullptr;
  unsigned* d_dense = nullptr;  // TableView::dense_flag
  i64* d_scalar = nullptr;
  i64* h_scalar = nullptr;  // pinned
  unsigned* progress_host = nullptr;  // tfra_table_step_prefetch: pinned progress counter of the main stream
  unsigned* own_stats_host = nullptr; // pinned: [0] keys that were NOT plain hits, [1] keys looked at — a sample (every 16th wave) of the last
                                      // ownership write-back of a SET plan that has ended: picks the pass's form for the next one (launch_own)
  unsigned step_gen = 0;
  std::mutex step_mu;
  uint8_t* evict_flags = nullptr;  // phase-2 flags of a fused write-back on a bounded table
  size_t evict_flags_cap = 0;
  int* winner = nullptr;
  size_t winner_len = 0;
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  void* own_plan = nullptr;  // tfra_sparse_plan of the one-call write-backs (tfra_table_apply_sparse / upsert_sparse)
  void* big_ws = nullptr;    // tfra_workspace of tfra_table_apply_sparse with more than 2^18 ids
  unsigned* own_tags = nullptr;  // [nb] bucket-owner tags (upsert_own_kernel), allocated on first use
  u64 own_tags_nb = 0;
  unsigned own_gen = 0;      // bucket-owner tag of the last ownership-based write-back (upsert_own_kernel)
  void* own_ws = nullptr;    // scratch of the ownership pass over a caller's unique keys (own_upsert_unique): counters | items | flags
  size_t own_ws_bytes = 0;
  unsigned own_ws_uses = 0;
  unsigned apply_P = 0;      // bucket count the cursor area at the head of `scratch` is armed for (0 = not armed)
  AuxInitPod aux{};
  // host bookkeeping
  std::mutex mu;
  size_t size_ub = 0;  // upper bound of the live-key count (exact after read_size)
  hipStream_t last_stream = nullptr;
  bool has_last = false;
  hipEvent_t chain_event = nullptr;
  hipEvent_t size_event = nullptr;  // async size refresh (prepare_insert)
  bool size_pending = false;
  size_t n_since_read = 0;
  i64* h_size = nullptr;  // pinned, inside the h_scalar block
  bool growth_blocked = false;
  bool dense = false;        // a table that cannot grow any more holds > 60 % of its slots (async size reads)
  bool capture_safe = false;  // TFRA_OPTION_CAPTURE_SAFE
  bool no_owner_tags = false; // TFRA_OPTION_NO_OWNER_TAGS
  int key_file_bytes = 8;     // TFRA_OPTION_KEY_BYTES_ON_DISK
  uint64_t global_epoch = 0;
  int64_t curr_step = 1;
  bool epoch_hold = false;   // see step_epoch (tfra_optim.hip)
  int n_rehash = 0;

  void* dalloc(size_t bytes, hipStream_t s);
  void dfree(void* p, hipStream_t s);
  int alloc_storage(u64 nb, Storage* st, hipStream_t s);
  TableView view_of(const Storage& st) const;
  int enter(hipStream_t s);
  int read_size(hipStream_t s, size_t* out);
  int check_errors(hipStream_t s);
  int ensure_winner(hipStream_t s);
  unsigned* ensure_own_tags(hipStream_t s);
  int ensure_scratch(size_t bytes, hipStream_t s);
  int grow(u64 min_nb, hipStream_t s);
  u64 lattice_nb(u64 min_nb) const;
  bool at_max_capacity() const;
  int grow_in_place(u64 min_nb, hipStream_t s);   // TFRA_ERR_UNSUPPORTED: not possible here, copy instead
  void free_storage(Storage& st, hipStream_t s);
  int n_split = 0;                                // in-place growths so far
  int prepare_insert(size_t n, hipStream_t s);
  int poll_density(size_t n, hipStream_t s);
  int bounded_flags(size_t n, hipStream_t s, uint8_t** out);
};

void destroy_own_plan(Table* t);   // tfra_csr.hip
// insert_or_assign of UNIQUE keys as one ownership pass (tfra_csr.hip); *taken = false: not applicable, run the locked kernels
int own_upsert_unique(Table* t, hipStream_t s, size_t n, const i64* keys, const void* values, const u64* scores, bool* taken,
                      const uint8_t* accum_exists = nullptr, const int64_t* d_n = nullptr);   // accum_exists: insert_or_accum instead of an assign
void destroy_workspace_plan(void* plan);   // tfra_csr.hip
void step_epoch_public(Table* t);  // tfra_optim.hip

}  // namespace tfra

// Scratch of the front-end ops (tfra_workspace_create): one growing device buffer per caller stream.
struct tfra_workspace {
  int device = 0;
  void* buf = nullptr;
  size_t bytes = 0;
  void* plan = nullptr;   // tfra_sparse_plan of tfra_reduce_by_key
  void* uplan = nullptr;  // tfra_sparse_plan of tfra_unique_unordered
  unsigned* h_err = nullptr;   // pinned: polls of the one-launch unique that timed out (reported by the NEXT tfra_unique_unordered call)
  // tfra_unique (up to 2^20 ids): two persistent hash sets that alternate and empty each other (no fill kernel per call)
  void* unq_buf = nullptr;
  size_t unq_cap = 0, unq_nmax = 0;
  unsigned unq_parity = 0, unq_gen = 0;
  // the sets persist from call to call: a call on ANOTHER stream than the previous one is ordered behind it by an event
  hipStream_t unq_stream = nullptr;
  bool unq_stream_set = false;
  hipEvent_t unq_ev = nullptr;
  int ensure(size_t need, hipStream_t s) {
    if (need <= bytes) return TFRA_OK;
    if (buf) {
      if (hipStreamSynchronize(s) != hipSuccess || hipFree(buf) != hipSuccess) return tfra::set_error(TFRA_ERR_HIP, "workspace: free failed");
      buf = nullptr; bytes = 0;
    }
    size_t want = need > ((size_t)1 << 20) ? need : ((size_t)1 << 20);
    hipError_t e = hipMalloc(&buf, want);
    if (e != hipSuccess) { buf = nullptr; return tfra::set_error(e == hipErrorOutOfMemory ? TFRA_ERR_OOM : TFRA_ERR_HIP, "workspace: hipMalloc failed"); }
    bytes = want;
    return TFRA_OK;
  }
};
