// TEST INFRASTRUCTURE ONLY — never linked into, imported by or executed from the product
// path (recommenders-addons_amd/).  Allowed callers: tests/, __graft_entry__.smoke(),
// bench.py's cpu_baseline leg.
//
// oracle/_ref/libtfra_ref.so = the reference's OWN CPU storage engine, compiled in place.
//
// This file is a thin raw-pointer host for the reference's vendored, header-only
// libcuckoo (incl. TFRA's `insert_or_accum` addition).  The header is #included from
// where it lies under /root/reference (see oracle/Makefile, -I flag); nothing of it is
// copied into this repository.  What IS restated here, because the originals need
// TensorFlow/Eigen headers that are absent from this image, is the ~100-line adapter the
// reference puts between TF tensors and cuckoohash_map:
//
//   R = /root/reference/tensorflow_recommenders_addons/dynamic_embedding/core
//   R/kernels/lookup_impl/lookup_table_op_cpu.h:42-52    ValueArray::operator+=  (accum order)
//   R/kernels/lookup_impl/lookup_table_op_cpu.h:53-63    DefaultValueArray::operator+=
//   R/kernels/lookup_impl/lookup_table_op_cpu.h:90-101   HybridHash<int64> (murmur3 fmix64)
//   R/kernels/lookup_impl/lookup_table_op_cpu.h:148-263  TableWrapperOptimized<K,V,DIM>
//   R/kernels/lookup_impl/lookup_table_op_cpu.h:265-389  TableWrapperDefault<K,V>
//   R/kernels/lookup_impl/lookup_table_op_cpu.h:403-412  dispatch rule (int64 key && DIM<=100)
//   R/kernels/cuckoo_hashtable_op.cc:39-182              LaunchTensors{Find,FindWithExists,
//                                                        Insert,Accum} (Shard fan-out,
//                                                        is_full_default rule :48-50)
//   R/kernels/cuckoo_hashtable_op.cc:268-308             Remove (serial), Clear, Export
//
// Differences from the reference adapter, all irrelevant to results:
//   * TTypes<V,2>::Tensor -> raw pointers + row stride = value_dim;
//   * tensorflow::Shard over the intra-op pool -> static contiguous split over
//     `threads` std::threads (threads<=1 runs inline, which is what parity tests use so
//     duplicate-key order is the sequential one);
//   * the DIM 1..100 macro fan-out is instantiated only for the dims listed in
//     REF_OPT_DIMS (float values); every other (V, dim) takes the TableWrapperDefault
//     twin, which the reference itself documents as result-identical.
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "cuckoohash_map.hh"  // -> /root/reference/.../core/lib/cuckoo (via -I)

namespace {

using i64 = long long;  // TF int64 is `long long` on Linux

// lookup_table_op_cpu.h:90-101
struct HybridHashI64 {
  inline std::size_t operator()(i64 const& key) const noexcept {
    uint64_t k = static_cast<uint64_t>(key);
    k ^= k >> 33;
    k *= UINT64_C(0xff51afd7ed558ccd);
    k ^= k >> 33;
    k *= UINT64_C(0xc4ceb9fe1a85ec53);
    k ^= k >> 33;
    return static_cast<std::size_t>(k);
  }
};

// lookup_table_op_cpu.h:42-52
template <class V, size_t DIM>
class ValueArray final : public std::array<V, DIM> {
 public:
  inline ValueArray<V, DIM>& operator+=(const ValueArray<V, DIM>& rhs) noexcept {
    for (size_t i = 0; i < DIM; i++) (*this)[i] += rhs[i];
    return *this;
  }
};

// lookup_table_op_cpu.h:53-63 (gtl::InlinedVector<V,2> -> std::vector<V>: same element
// order, same `a = a + b` per element)
template <class V>
class DefaultValueArray final : public std::vector<V> {
 public:
  inline DefaultValueArray<V>& operator+=(const DefaultValueArray<V>& rhs) noexcept {
    for (size_t i = 0; i < this->size(); i++) (*this)[i] = ((*this)[i]) + rhs[i];
    return *this;
  }
};

struct TableBase {
  virtual ~TableBase() {}
  virtual bool insert_or_assign(i64 key, const void* values, i64 dim, i64 index) = 0;
  virtual bool insert_or_accum(i64 key, const void* vod, bool exist, i64 dim, i64 index) = 0;
  virtual bool find(i64 key, void* values, const void* defaults, i64 dim, bool full,
                    i64 index) = 0;
  virtual size_t dump(i64* keys, void* values, size_t off, size_t len) = 0;
  virtual size_t size() = 0;
  virtual void clear() = 0;
  virtual bool erase(i64 key) = 0;
  virtual size_t elem_size() const = 0;
};

// lookup_table_op_cpu.h:148-263
template <class V, size_t DIM>
class TableOptimized final : public TableBase {
  using ValueType = ValueArray<V, DIM>;
  using Table = cuckoohash_map<i64, ValueType, HybridHashI64>;

 public:
  explicit TableOptimized(size_t init_size) : table_(new Table(init_size)) {}
  bool insert_or_assign(i64 key, const void* values, i64 dim, i64 index) override {
    ValueType v;
    std::copy_n(static_cast<const V*>(values) + index * dim, dim, v.begin());
    return table_->insert_or_assign(key, v);
  }
  bool insert_or_accum(i64 key, const void* vod, bool exist, i64 dim, i64 index) override {
    ValueType v;
    std::copy_n(static_cast<const V*>(vod) + index * dim, dim, v.begin());
    return table_->insert_or_accum(key, v, exist);
  }
  bool find(i64 key, void* values, const void* defaults, i64 dim, bool full,
            i64 index) override {
    ValueType v;
    V* out = static_cast<V*>(values) + index * dim;
    const V* def = static_cast<const V*>(defaults);
    bool exist = table_->find(key, v);
    if (exist) {
      std::copy_n(v.begin(), dim, out);
    } else {
      for (i64 j = 0; j < dim; j++) out[j] = full ? def[index * dim + j] : def[j];
    }
    return exist;
  }
  size_t dump(i64* keys, void* values, size_t off, size_t len) override {
    auto lt = table_->lock_table();
    auto lt_size = lt.size();
    if (off > lt_size || lt_size == 0) return 0;
    auto b = lt.begin();
    for (size_t i = 0; i < off; ++i) ++b;
    auto e = b;
    if (off + len >= lt_size) {
      e = lt.end();
    } else {
      for (size_t i = 0; i < len; ++i) ++e;
    }
    V* val_ptr = static_cast<V*>(values);
    size_t n = 0;
    for (auto it = b; it != e; ++it, ++keys, val_ptr += DIM) {
      *keys = it->first;
      std::copy_n(it->second.begin(), DIM, val_ptr);
      ++n;
    }
    return n;
  }
  size_t size() override { return table_->size(); }
  void clear() override { table_->clear(); }
  bool erase(i64 key) override { return table_->erase(key); }
  size_t elem_size() const override { return sizeof(V); }

 private:
  std::unique_ptr<Table> table_;
};

// lookup_table_op_cpu.h:265-389
template <class V>
class TableDefault final : public TableBase {
  using ValueType = DefaultValueArray<V>;
  using Table = cuckoohash_map<i64, ValueType, HybridHashI64>;

 public:
  explicit TableDefault(size_t init_size) : table_(new Table(init_size)) {}
  static ValueType load(const void* src, i64 dim, i64 index) {
    ValueType v;
    v.reserve(dim);
    const V* p = static_cast<const V*>(src) + index * dim;
    for (i64 j = 0; j < dim; j++) v.push_back(p[j]);
    return v;
  }
  bool insert_or_assign(i64 key, const void* values, i64 dim, i64 index) override {
    return table_->insert_or_assign(key, load(values, dim, index));
  }
  bool insert_or_accum(i64 key, const void* vod, bool exist, i64 dim, i64 index) override {
    return table_->insert_or_accum(key, load(vod, dim, index), exist);
  }
  bool find(i64 key, void* values, const void* defaults, i64 dim, bool full,
            i64 index) override {
    ValueType v;
    v.reserve(dim);
    V* out = static_cast<V*>(values) + index * dim;
    const V* def = static_cast<const V*>(defaults);
    bool exist = table_->find(key, v);
    if (exist) {
      std::copy_n(v.begin(), dim, out);
    } else {
      for (i64 j = 0; j < dim; j++) out[j] = full ? def[index * dim + j] : def[j];
    }
    return exist;
  }
  size_t dump(i64* keys, void* values, size_t off, size_t len) override {
    auto lt = table_->lock_table();
    auto lt_size = lt.size();
    if (off > lt_size || lt_size == 0) return 0;
    auto b = lt.begin();
    for (size_t i = 0; i < off; ++i) ++b;
    auto e = b;
    if (off + len >= lt_size) {
      e = lt.end();
    } else {
      for (size_t i = 0; i < len; ++i) ++e;
    }
    const auto dim = (lt.begin()->second).size();
    V* val_ptr = static_cast<V*>(values);
    size_t n = 0;
    for (auto it = b; it != e; ++it, ++keys, val_ptr += dim) {
      *keys = it->first;
      std::copy_n(it->second.begin(), dim, val_ptr);
      ++n;
    }
    return n;
  }
  size_t size() override { return table_->size(); }
  void clear() override { table_->clear(); }
  bool erase(i64 key) override { return table_->erase(key); }
  size_t elem_size() const override { return sizeof(V); }

 private:
  std::unique_ptr<Table> table_;
};

struct Handle {
  std::unique_ptr<TableBase> t;
  i64 dim;
};

template <class V>
TableBase* make_default(size_t init) {
  return new TableDefault<V>(init);
}

// The reference fans DIM out over 1..100 (lookup_table_op_cpu.h:414-468); we instantiate the
// dims the baselines/KATs use for float and let everything else take the Default twin.
#define REF_OPT_DIMS(X) X(1) X(2) X(4) X(8) X(10) X(16) X(32) X(64) X(100)

TableBase* make_table(int dtype, i64 dim, size_t init) {
  switch (dtype) {
    case 0:  // float32
#define X(D) \
  if (dim == D) return new TableOptimized<float, D>(init);
      REF_OPT_DIMS(X)
#undef X
      return make_default<float>(init);
    case 3:
      return make_default<int8_t>(init);
    case 4:
      return make_default<int32_t>(init);
    case 5:
      return make_default<i64>(init);
    case 6:
      return make_default<double>(init);
    default:
      return nullptr;  // half / bfloat16 need Eigen's types: not hostable here
  }
}

// tensorflow::Shard equivalent (cuckoo_hashtable_op.cc:62-64).  The reference hands Shard a cost per key of
// (#elements / #threads + 1), i.e. "expensive": TF's cost model then cuts the keys into ~#threads contiguous
// blocks and runs them on the device's PERSISTENT intra-op pool.  Same here: one process-wide pool per thread
// count, created on first use, workers spin briefly between jobs (like Eigen's pool) and sleep after 2 ms idle,
// static contiguous split.  (Spawning std::threads per op, as this shim first did, cost more than the op.)
class Pool {
 public:
  explicit Pool(int n) : n_(n) {
    for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t] { run(t); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; gen_.fetch_add(1); }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  template <class F>
  void parallel(i64 total, F fn) {
    const i64 per = (total + n_ - 1) / n_;
    job_ = [=](int t) {
      const i64 b = t * per, e = std::min<i64>(total, b + per);
      if (b < e) fn(b, e);
    };
    pending_.store(n_ - 1, std::memory_order_relaxed);
    { std::lock_guard<std::mutex> l(mu_); gen_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    job_(0);
    while (pending_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  }

 private:
  void run(int t) {
    unsigned long seen = 0;
    for (;;) {
      // spin ~2 ms, then block
      auto t0 = std::chrono::steady_clock::now();
      while (gen_.load(std::memory_order_acquire) == seen) {
        __builtin_ia32_pause();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
          std::unique_lock<std::mutex> l(mu_);
          cv_.wait(l, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        }
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_) return;
      job_(t);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  int n_;
  std::vector<std::thread> workers_;
  std::function<void(int)> job_;
  std::atomic<unsigned long> gen_{0};
  std::atomic<int> pending_{0};
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false;
};

std::mutex g_pools_mu;
std::map<int, std::unique_ptr<Pool>> g_pools;

template <class F>
void shard(int threads, i64 total, F fn) {
  if (threads <= 1 || total < 2 * threads) {
    fn(0, total);
    return;
  }
  Pool* p;
  {
    std::lock_guard<std::mutex> l(g_pools_mu);
    auto& slot = g_pools[threads];
    if (!slot) slot.reset(new Pool(threads));
    p = slot.get();
  }
  p->parallel(total, fn);
}

}  // namespace

extern "C" {

// dtype codes follow include/tfra_mi355x.h (0 f32, 3 i8, 4 i32, 5 i64, 6 f64).
void* tfra_ref_create(int dtype, long long dim, unsigned long long init_size) {
  // cuckoo_hashtable_op.cc:199-207: init_size 0 -> TF_HASHTABLE_INIT_SIZE -> 8192
  if (init_size == 0) init_size = 1024 * 8;
  TableBase* t = make_table(dtype, dim, init_size);
  if (!t) return nullptr;
  Handle* h = new Handle;
  h->t.reset(t);
  h->dim = dim;
  return h;
}

void tfra_ref_destroy(void* hp) { delete static_cast<Handle*>(hp); }

// LaunchTensorsFind / FindWithExists (cuckoo_hashtable_op.cc:39-106). `exists` may be null.
void tfra_ref_find(void* hp, long long n, const long long* keys, void* values,
                   const void* defaults, int is_full_default, unsigned char* exists,
                   int threads) {
  Handle* h = static_cast<Handle*>(hp);
  shard(threads, n, [=](i64 b, i64 e) {
    for (i64 i = b; i < e; ++i) {
      bool ex = h->t->find(keys[i], values, defaults, h->dim, is_full_default != 0, i);
      if (exists) exists[i] = ex ? 1 : 0;
    }
  });
}

// LaunchTensorsInsert (cuckoo_hashtable_op.cc:111-150); clear!=0 is ImportValues (:288-291).
void tfra_ref_insert(void* hp, long long n, const long long* keys, const void* values,
                     int clear, int threads) {
  Handle* h = static_cast<Handle*>(hp);
  if (clear) h->t->clear();
  shard(threads, n, [=](i64 b, i64 e) {
    for (i64 i = b; i < e; ++i) h->t->insert_or_assign(keys[i], values, h->dim, i);
  });
}

// LaunchTensorsAccum (cuckoo_hashtable_op.cc:155-182)
void tfra_ref_accum(void* hp, long long n, const long long* keys, const void* vod,
                    const unsigned char* exists, int threads) {
  Handle* h = static_cast<Handle*>(hp);
  shard(threads, n, [=](i64 b, i64 e) {
    for (i64 i = b; i < e; ++i)
      h->t->insert_or_accum(keys[i], vod, exists[i] != 0, h->dim, i);
  });
}

// Remove is a serial loop in the reference (cuckoo_hashtable_op.cc:268-276)
void tfra_ref_remove(void* hp, long long n, const long long* keys) {
  Handle* h = static_cast<Handle*>(hp);
  for (i64 i = 0; i < n; ++i) h->t->erase(keys[i]);
}

void tfra_ref_clear(void* hp) { static_cast<Handle*>(hp)->t->clear(); }

// drops the persistent pools (threads exit); they are re-created on demand
void tfra_ref_pools_shutdown(void) {
  std::lock_guard<std::mutex> l(g_pools_mu);
  g_pools.clear();
}

unsigned long long tfra_ref_size(void* hp) { return static_cast<Handle*>(hp)->t->size(); }

// ExportValues -> dump(keys, values, 0, size) (cuckoo_hashtable_op.cc:293-308); the chunked
// form is what SaveToFileSystemImpl uses (:310-391).
unsigned long long tfra_ref_dump(void* hp, long long* keys, void* values,
                                 unsigned long long offset, unsigned long long length) {
  return static_cast<Handle*>(hp)->t->dump(keys, values, offset, length);
}

}  // extern "C"
