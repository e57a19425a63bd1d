// Many tables per GPU (BASELINE configs[4]: 26 embedding tables of a DLRM-style model): one C call issues the training
// step of every table — tfra_table_step_prefetch / _assign each — from a small pool of host threads, one group of tables
// per thread, each table on its own pair of streams.  A step of one table is 6-8 kernel launches (~5 us of host time
// each on ROCm 7.2): issued one table after the other from one thread, 26 tables are host-bound (1.9 ms per step measured
// in round 1, the GPU idle most of the time); issued from W threads onto disjoint streams the launches overlap and so do
// the kernels of different tables.  The reference has no counterpart: TensorFlow's executor runs the per-table op
// sequences of one session.run on its inter-op thread pool, which is what this call stands in for.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tfra_mi355x.h"
#include "tfra_host.h"

using namespace tfra;

namespace {

struct Pool {
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  const tfra_step_desc* descs = nullptr;
  size_t n = 0;
  int device = 0;
  unsigned generation = 0;     // bumped per call
  size_t next = 0;             // next desc to take
  size_t finished = 0;
  int active_workers = 0;      // workers that take part in the current call
  int rc = TFRA_OK;
  std::string err;
  bool stop = false;

  void worker(int id) {
    unsigned seen = 0;
    int dev_set = -1;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv_work.wait(lk, [&] { return stop || (generation != seen && id < active_workers); });
      if (stop) return;
      seen = generation;
      const int dev = device;
      while (next < n) {
        const size_t i = next++;
        const tfra_step_desc d = descs[i];
        lk.unlock();
        if (dev_set != dev) { (void)hipSetDevice(dev); dev_set = dev; }
        int r;
        if (d.opt)
          r = tfra_table_step_prefetch(d.table, d.opt, d.plan_cur, d.ids_cur, d.rows_out, d.find_default, (const float*)d.grads_or_values,
                                       d.param_default_row, d.plan_next, d.ids_next, d.n_next, d.main_stream, d.side_stream);
        else
          r = tfra_table_step_prefetch_assign(d.table, d.plan_cur, d.ids_cur, d.rows_out, d.find_default, d.grads_or_values, d.scores,
                                              d.plan_next, d.ids_next, d.n_next, d.main_stream, d.side_stream);
        std::string e = r ? std::string(tfra_last_error()) : std::string();
        lk.lock();
        if (r && rc == TFRA_OK) { rc = r; err = "table " + std::to_string(i) + ": " + e; }
        ++finished;
      }
      if (finished == n) cv_done.notify_all();
    }
  }

  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : threads) t.join();
  }
};

Pool& pool() {
  static Pool p;
  return p;
}
std::mutex g_call_mu;   // one multi-table call at a time per process

}  // namespace

extern "C" int tfra_multi_step_prefetch(size_t n_tables, const tfra_step_desc* descs, int n_workers) {
  if (n_tables == 0) return TFRA_OK;
  if (!descs) return set_error(TFRA_ERR_INVALID, "multi_step: null descriptors");
  for (size_t i = 0; i < n_tables; ++i)
    if (descs[i].struct_size != sizeof(tfra_step_desc)) return set_error(TFRA_ERR_INVALID, "multi_step: descriptor size mismatch");
  if (n_workers <= 0) n_workers = 4;
  if ((size_t)n_workers > n_tables) n_workers = (int)n_tables;
  if (n_workers > 32) n_workers = 32;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return set_error(TFRA_ERR_HIP, "multi_step: no device");
  std::lock_guard<std::mutex> call_lock(g_call_mu);
  Pool& p = pool();
  std::unique_lock<std::mutex> lk(p.mu);
  while ((int)p.threads.size() < n_workers) {
    const int id = (int)p.threads.size();
    p.threads.emplace_back([&p, id] { p.worker(id); });
  }
  p.descs = descs; p.n = n_tables; p.device = dev; p.next = 0; p.finished = 0; p.rc = TFRA_OK; p.err.clear();
  p.active_workers = n_workers;
  ++p.generation;
  p.cv_work.notify_all();
  p.cv_done.wait(lk, [&] { return p.finished == p.n; });
  p.active_workers = 0;
  if (p.rc) return set_error(p.rc, "multi_step: " + p.err);
  return TFRA_OK;
}
