// KV-file save / load for the MI355X table (N2 row of SURVEY.md §8f).
//
// On-disk format = the reference's, byte for byte: `<prefix>-keys` is a raw native-endian
// K array (int64; int32 for tables whose op-level key type is int32: TFRA_OPTION_KEY_BYTES_ON_DISK = 4, narrowed / widened on
// the host side of the staging buffers), `<prefix>-values` a raw V[n*dim] array in the same order
// (R/kernels/cuckoo_hashtable_op.cc:310-391 SaveToFileSystemImpl, :393-505 LoadFromFileSystemImpl;
// GPU twin R/kernels/lookup_impl/lookup_table_op_hkv.h:132-273 RandomKVFile, :602-717).
// Streaming: `buffer_keys` slots are exported per chunk through pinned staging buffers, so a
// 288-GB table never needs a second device copy.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/tfra_mi355x.h"
#include "tfra_device.h"
#include "tfra_host.h"

using namespace tfra;

namespace {
struct File {
  FILE* f = nullptr;
  ~File() { if (f) fclose(f); }
};
}  // namespace

extern "C" {

static int save_impl(tfra_table_t* tp, int field, const char* prefix, size_t buffer_keys, int append, tfra_stream_t stream,
                     size_t* n_saved) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !prefix) return set_error(TFRA_ERR_INVALID, "save: null argument");
  if (field < 0 || field > t->opts.aux_fields) return set_error(TFRA_ERR_INVALID, "save: bad field");
  if (buffer_keys == 0) buffer_keys = 4194304;  // Python default buffer_size
  hipStream_t s = (hipStream_t)stream;
  size_t cap = 0;
  int rc = tfra_table_capacity(tp, &cap);
  if (rc) return rc;
  const size_t fb = t->field_bytes;
  std::string kp = std::string(prefix) + "-keys", vp = std::string(prefix) + "-values";
  // write to temp files then rename (the reference does this when the FS has no atomic move)
  std::string kt = append ? kp : kp + ".tmp", vt = append ? vp : vp + ".tmp";
  File kf, vf;
  kf.f = fopen(kt.c_str(), append ? "ab" : "wb");
  vf.f = fopen(vt.c_str(), append ? "ab" : "wb");
  if (!kf.f || !vf.f) return set_error(TFRA_ERR_IO, "save: cannot open " + kt + " / " + vt);
  size_t chunk = std::min(buffer_keys, cap);
  i64 *d_keys = nullptr, *h_keys = nullptr;
  unsigned char *d_vals = nullptr, *h_vals = nullptr;
  size_t *d_cnt = nullptr, *h_cnt = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(d_keys); (void)hipFree(d_vals); (void)hipFree(d_cnt);
    (void)hipHostFree(h_keys); (void)hipHostFree(h_vals); (void)hipHostFree(h_cnt);
  };
  if (hipMalloc((void**)&d_keys, chunk * sizeof(i64)) != hipSuccess || hipMalloc((void**)&d_vals, chunk * fb) != hipSuccess ||
      hipMalloc((void**)&d_cnt, sizeof(size_t)) != hipSuccess || hipHostMalloc((void**)&h_keys, chunk * sizeof(i64)) != hipSuccess ||
      hipHostMalloc((void**)&h_vals, chunk * fb) != hipSuccess || hipHostMalloc((void**)&h_cnt, sizeof(size_t)) != hipSuccess) {
    cleanup();
    return set_error(TFRA_ERR_OOM, "save: staging allocation failed");
  }
  size_t total = 0;
  for (size_t off = 0; off < cap; off += chunk) {
    size_t len = std::min(chunk, cap - off);
    if (hipMemsetAsync(d_cnt, 0, sizeof(size_t), s) != hipSuccess) { cleanup(); return set_error(TFRA_ERR_HIP, "save: memset"); }
    rc = tfra_table_export_batch(tp, len, off, d_cnt, (int64_t*)d_keys, field == 0 ? d_vals : nullptr, nullptr, stream);
    if (rc) { cleanup(); return rc; }
    (void)hipMemcpyAsync(h_cnt, d_cnt, sizeof(size_t), hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cleanup(); return set_error(TFRA_ERR_HIP, "save: sync"); }
    size_t got = *h_cnt;
    if (!got) continue;
    if (field > 0) {   // a co-located state vector (optimizer slot): read it for the exported keys, same order
      // every exported key is resident, so the default row is never used: any valid row-sized buffer will do
      rc = tfra_table_find_field(tp, field, got, (const int64_t*)d_keys, d_vals, nullptr, d_vals, 1, stream);
      if (rc) { cleanup(); return rc; }
    }
    (void)hipMemcpyAsync(h_keys, d_keys, got * sizeof(i64), hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(h_vals, d_vals, got * fb, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cleanup(); return set_error(TFRA_ERR_HIP, "save: copy"); }
    if (t->key_file_bytes == 4) {   // int32 keys on disk: narrowed in place (the staging buffer is ours)
      int* k32 = reinterpret_cast<int*>(h_keys);
      for (size_t i = 0; i < got; ++i) {
        const i64 k = h_keys[i];
        if ((i64)(int)k != k) { cleanup(); return set_error(TFRA_ERR_INVALID, "save: a key does not fit the table's 4-byte key files"); }
        k32[i] = (int)k;
      }
    }
    if (fwrite(h_keys, (size_t)t->key_file_bytes, got, kf.f) != got || fwrite(h_vals, fb, got, vf.f) != got) {
      cleanup();
      return set_error(TFRA_ERR_IO, "save: short write");
    }
    total += got;
  }
  cleanup();
  fclose(kf.f); kf.f = nullptr;
  fclose(vf.f); vf.f = nullptr;
  if (!append && (rename(kt.c_str(), kp.c_str()) != 0 || rename(vt.c_str(), vp.c_str()) != 0))
    return set_error(TFRA_ERR_IO, "save: rename failed");
  if (n_saved) *n_saved = total;
  return TFRA_OK;
}

int tfra_table_save(tfra_table_t* tp, const char* prefix, size_t buffer_keys, int append, tfra_stream_t stream,
                    size_t* n_saved) {
  return save_impl(tp, 0, prefix, buffer_keys, append, stream, n_saved);
}

int tfra_table_save_field(tfra_table_t* tp, int field, const char* prefix, size_t buffer_keys, int append,
                          tfra_stream_t stream, size_t* n_saved) {
  return save_impl(tp, field, prefix, buffer_keys, append, stream, n_saved);
}

static int load_impl(tfra_table_t* tp, int field, const char* prefix, size_t buffer_keys, tfra_stream_t stream, size_t* n_loaded) {
  Table* t = reinterpret_cast<Table*>(tp);
  if (!t || !prefix) return set_error(TFRA_ERR_INVALID, "load: null argument");
  if (field < 0 || field > t->opts.aux_fields) return set_error(TFRA_ERR_INVALID, "load: bad field");
  if (buffer_keys == 0) buffer_keys = 4194304;
  hipStream_t s = (hipStream_t)stream;
  const size_t fb = t->field_bytes;
  std::string kp = std::string(prefix) + "-keys", vp = std::string(prefix) + "-values";
  File kf, vf;
  kf.f = fopen(kp.c_str(), "rb");
  vf.f = fopen(vp.c_str(), "rb");
  if (!kf.f || !vf.f) return set_error(TFRA_ERR_IO, "load: cannot open " + kp + " / " + vp);
  fseek(kf.f, 0, SEEK_END);
  size_t key_bytes = (size_t)ftell(kf.f);
  fseek(kf.f, 0, SEEK_SET);
  fseek(vf.f, 0, SEEK_END);
  size_t val_bytes = (size_t)ftell(vf.f);
  fseek(vf.f, 0, SEEK_SET);
  const size_t kb = (size_t)t->key_file_bytes;
  size_t nkeys = key_bytes / kb;
  // LoadFromFileSystemImpl checks the two files describe the same number of entries (:431-441)
  if (key_bytes % kb || nkeys * fb != val_bytes)
    return set_error(TFRA_ERR_IO, "load: " + kp + " and " + vp + " sizes do not match dim");
  size_t chunk = std::max<size_t>(1, std::min(buffer_keys, nkeys));
  i64 *d_keys = nullptr, *h_keys = nullptr;
  unsigned char *d_vals = nullptr, *h_vals = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(d_keys); (void)hipFree(d_vals); (void)hipHostFree(h_keys); (void)hipHostFree(h_vals);
  };
  if (hipMalloc((void**)&d_keys, chunk * sizeof(i64)) != hipSuccess || hipMalloc((void**)&d_vals, chunk * fb) != hipSuccess ||
      hipHostMalloc((void**)&h_keys, chunk * sizeof(i64)) != hipSuccess || hipHostMalloc((void**)&h_vals, chunk * fb) != hipSuccess) {
    cleanup();
    return set_error(TFRA_ERR_OOM, "load: staging allocation failed");
  }
  size_t done = 0;
  while (done < nkeys) {
    size_t len = std::min(chunk, nkeys - done);
    if (fread(h_keys, kb, len, kf.f) != len || fread(h_vals, fb, len, vf.f) != len) {
      cleanup();
      return set_error(TFRA_ERR_IO, "load: short read");
    }
    if (kb == 4) {   // int32 keys on disk: widened in place, back to front
      const int* k32 = reinterpret_cast<const int*>(h_keys);
      for (size_t i = len; i-- > 0;) h_keys[i] = (i64)k32[i];
    }
    (void)hipMemcpyAsync(d_keys, h_keys, len * sizeof(i64), hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(d_vals, h_vals, len * fb, hipMemcpyHostToDevice, s);
    // Files written by save hold unique keys; files concatenated by hand may not, so a growing table takes the
    // duplicate-safe path (last one wins).  A bounded (Hkv) table at max_capacity needs one writer per key for its
    // eviction (HKV's contract, tfra_table_insert_or_assign), so there the keys of a chunk are taken as unique.
    const uint32_t flags = (t->opts.strategy >= 0 && t->opts.max_capacity) ? TFRA_FLAG_UNIQUE_KEYS : 0u;
    int rc = field == 0 ? tfra_table_insert_or_assign(tp, len, (const int64_t*)d_keys, d_vals, nullptr, flags, stream)
                        : tfra_table_insert_field(tp, field, len, (const int64_t*)d_keys, d_vals, flags, stream);
    if (rc) { cleanup(); return rc; }
    if (hipStreamSynchronize(s) != hipSuccess) { cleanup(); return set_error(TFRA_ERR_HIP, "load: sync"); }
    done += len;
  }
  cleanup();
  if (n_loaded) *n_loaded = done;
  return TFRA_OK;
}

int tfra_table_load(tfra_table_t* tp, const char* prefix, size_t buffer_keys, tfra_stream_t stream, size_t* n_loaded) {
  return load_impl(tp, 0, prefix, buffer_keys, stream, n_loaded);
}

int tfra_table_load_field(tfra_table_t* tp, int field, const char* prefix, size_t buffer_keys, tfra_stream_t stream,
                          size_t* n_loaded) {
  return load_impl(tp, field, prefix, buffer_keys, stream, n_loaded);
}

}  // extern "C"
