"""bench.py — dynamic-embedding hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config m1b|c3|c2|c4|c5] [--slots S] [--keys N_KEYS] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic ids.  Configurations (BASELINE.json `configs`):

  m1b (default at N=1: the configuration BASELINE.json's `metric` is quoted on) dim=64 fp32, a 10^9-slot bounded (Hkv, LRU)
      table pre-filled to capacity, batch B=131072 of Zipf-1.2 ids over the 10^9 ranks:
          lookup(B ids, misses get the default row)  ->  insert_or_assign(B ids, B rows; repeats: last one wins)
      If 10^9 slots do not allocate, the largest slot count that does is used and named in config.workload.
  c3  configs[2]: the same table shape with dim=128 fp16 rows (256 B, the same row bytes) and 50 % never-seen ids per
      batch (monotone counter): the timed region carries inserts AND score-based eviction.
  c2  configs[1]: growing table, 100 M resident keys, dim=64 fp32 rows [p|m|v], Zipf-1.2 batch, `--new-key-ratio`
      (default 0.1) of every batch are never-seen keys:
          lookup(B) -> gradients [B,64] -> duplicate ids summed -> fused sparse Adam on the unique keys
  c4  configs[3], the per-GPU workload of EVERY `--gpus N` run with N > 1 (and of `--config c4 --gpus 1`): N shards of
      5*10^8 keys each, hash-sharded by the reference's default partitioner, dim=64 fp32, per-GPU batch B drawn from the
      GLOBAL Zipf-1.2 over the N*5*10^8 keys, ids / rows / gradients routed by tfra_route_* (RCCL alltoall over xGMI; at
      N=1 the same driver without a transport):  routed lookup(B) -> routed fused SGD write-back of the batch's keys.
  c5  configs[4]: 26 tables, dims {16,32,64,128}, fused FTRL.  `--config c5` on one GPU: the local multi-table drivers
      (tfra_multi_step_prefetch ...) plus the N = 1 point of the routed form; `--config c5 --gpus N`: every table hash-sharded over
      the N ranks, one tfra_route per table on one shared transport (MultiTableRoutedStep), per-GPU batch B per table.
  The default invocation at N=1 prints m1b as the top-level line and c3 / c2 / c4 / c5 (routed, small tables) under "secondary".

`value` = ids looked up AND written back per second (lookup+insert pairs/s), whole job.  On the assign workloads (m1b, c3) the
faster of the table's two step drivers, both reported (`faster_driver` names it):
  `value_overlapped_step`   tfra_table_steps_overlap: ONE launch per step = lookup of batch i+1 (ids of batch i served from the rows
                            being written) + write-back of batch i + the plans of the next two batches; ids known two batches ahead;
  `value_look_ahead_driver` tfra_table_step_prefetch_assign: one C call per step, the plan of batch i+1 on a second HIP stream;
on the gradient workloads (c2, c4) the look-ahead driver with the fused optimizer.  Next to it, without any look-ahead:
  `value_plain_call`  tfra_table_find + tfra_table_upsert_sparse — the second call is a fused extra that de-duplicates
                      on the device inside the call (NOT an op of the reference's surface);
  `value_op_surface`  the reference's op order (python/ops/dynamic_embedding_ops.py:99-117): tfra_unique_unordered -> Find(U) ->
                      gather(B) -> Insert(U), Find and Insert reading the count on the device (enqueued before the ONE host read of
                      it: tf.unique's output shape); `value_op_surface_host_read_first`: the same with the read in front of Find —
                      exactly what the TF shim (tf_ops/mi355x_table_ops.h) issues; `value_op_surface_find_first`: round 3's order;
  `value_op_surface_table_ops_only` Find(B) + Insert(U) with the unique keys prepared beforehand;
  `value_accum`       bp_v2: Find(B) + insert_or_accum of the U unique keys.
Timing: after W warm-up steps, R = 5 back-to-back windows of exactly K steps each, every window bracketed by
barrier + torch.cuda.synchronize() (max over ranks); `value` / `ms_per_step` are the MEDIAN window, min / max are in
config.timing.  Inputs (id batches, values, gradients) are generated on the device and resident in HBM before the timed
region; an untimed verification pass checks the table against the values written (config.verified).
N>1: one process per GPU.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "recommenders-addons_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

ZIPF_S = 1.2
SEED = 20250205
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured for a plain copy)


# ------------------------------------------------------------------ synthetic inputs
def zipf_bounded(rng, size, n, s=ZIPF_S):
  """Bounded Zipf(s) over ranks 1..n by rejection-inversion (Hormann & Derflinger 1996)."""
  one_s = 1.0 - s

  def h_int(x):
    return (np.power(x, one_s) - 1.0) / one_s

  def h_int_inv(y):
    return np.power(1.0 + y * one_s, 1.0 / one_s)

  def h(x):
    return np.power(x, -s)

  hx1 = h_int(1.5) - 1.0
  hn = h_int(n + 0.5)
  sc = 2.0 - h_int_inv(h_int(2.5) - h(2.0))
  out = np.empty(size, dtype=np.int64)
  todo = np.arange(size)
  while todo.size:
    u = hn + rng.random(todo.size) * (hx1 - hn)
    x = h_int_inv(u)
    k = np.clip(np.floor(x + 0.5), 1, n)
    ok = (k - x <= sc) | (u >= h_int(k + 0.5) - h(k))
    out[todo[ok]] = k[ok].astype(np.int64)
    todo = todo[~ok]
  return out


def fmix64_np(x):
  x = x.astype(np.uint64)
  x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd)
  x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53)
  x ^= x >> np.uint64(33)
  return x


def keys_of_ranks(ranks):
  """Bijective scramble rank -> int64 key so hot keys are not adjacent."""
  with np.errstate(over="ignore"):
    return fmix64_np(ranks.astype(np.uint64) ^ np.uint64(SEED)).view(np.int64)


def keys_of_ranks_torch(torch, ranks):
  """Same scramble on the device (int64 two's-complement arithmetic, logical shifts emulated)."""
  def lsr33(v):
    return (v >> 33) & 0x7FFFFFFF
  x = ranks ^ SEED
  x = x ^ lsr33(x)
  x = x * (-49064778989728563)        # 0xff51afd7ed558ccd as int64
  x = x ^ lsr33(x)
  x = x * (-4265267296055464877)      # 0xc4ceb9fe1a85ec53 as int64
  x = x ^ lsr33(x)
  return x


def mixed_batches(rng, nbatch, batch, n_resident, new_ratio, fresh_start):
  """[nbatch, batch] ranks: a Zipf-1.2 draw over the resident ranks with `new_ratio` of the positions (evenly
  interleaved) replaced by never-seen ranks fresh_start, fresh_start+1, ... (monotone counter)."""
  ranks = zipf_bounded(rng, nbatch * batch, n_resident).reshape(nbatch, batch)
  n_new = int(round(batch * new_ratio))
  if n_new:
    pos = (np.arange(n_new) * (batch / n_new)).astype(np.int64)
    fresh = fresh_start + np.arange(nbatch * n_new, dtype=np.int64).reshape(nbatch, n_new)
    ranks[:, pos] = fresh
  return ranks, fresh_start + nbatch * n_new


def zipf_bounded_torch(torch, gen, size, n, dev, s=ZIPF_S):
  """zipf_bounded on the device (float64, the same rejection-inversion): a timed region of R x K steps needs hundreds of
  batches, one numpy draw of that many samples would take longer than the benchmark."""
  one_s = 1.0 - s
  h_int = lambda x: (torch.pow(x, one_s) - 1.0) / one_s
  h_int_inv = lambda y: torch.pow(1.0 + y * one_s, 1.0 / one_s)
  h = lambda x: torch.pow(x, -s)
  f64 = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)
  hx1 = h_int(f64(1.5)) - 1.0
  hn = h_int(f64(n + 0.5))
  sc = 2.0 - h_int_inv(h_int(f64(2.5)) - h(f64(2.0)))
  out = torch.empty(size, dtype=torch.int64, device=dev)
  todo = torch.arange(size, device=dev)
  while todo.numel():
    u = hn + torch.rand(todo.numel(), generator=gen, device=dev, dtype=torch.float64) * (hx1 - hn)
    x = h_int_inv(u)
    k = torch.clamp(torch.floor(x + 0.5), 1, n)
    ok = (k - x <= sc) | (u >= h_int(k + 0.5) - h(k))
    out[todo[ok]] = k[ok].to(torch.int64)
    todo = todo[~ok]
  return out


class IdFactory:
  """Batches of ids on the device: Zipf-1.2 over the ranks 1..n_resident, `new_ratio` of the positions (evenly interleaved)
  replaced by never-seen ranks from a monotone counter; rank -> key by the bijective scramble."""

  def __init__(self, torch, dev, batch, n_resident, new_ratio, fresh_start, seed):
    self.torch, self.dev, self.B, self.n, self.ratio = torch, dev, batch, n_resident, new_ratio
    self.fresh = fresh_start
    self.gen = torch.Generator(device=dev).manual_seed(seed)
    self.n_new = int(round(batch * new_ratio))
    self.pos = (torch.arange(self.n_new, device=dev, dtype=torch.float64) * (batch / max(1, self.n_new))).to(torch.int64)

  def ranks(self, nbatch):
    t = self.torch
    out = []
    for lo in range(0, nbatch, 64):   # bounded scratch: 64 batches per draw
      m = min(64, nbatch - lo)
      r = zipf_bounded_torch(t, self.gen, m * self.B, self.n, self.dev).reshape(m, self.B)
      if self.n_new:
        fresh = self.fresh + t.arange(m * self.n_new, device=self.dev, dtype=t.int64).reshape(m, self.n_new)
        r[:, self.pos] = fresh
        self.fresh += m * self.n_new
      out.append(r)
    return t.cat(out) if len(out) > 1 else out[0]

  def keys(self, nbatch):
    return keys_of_ranks_torch(self.torch, self.ranks(nbatch))


# ------------------------------------------------------------------ CPU baseline (reference engine)
def _cpu_baseline_worker(conn, batch, n_keys, threads, kind, growth_leg, init_div=1):
  """Child process (no torch): builds the reference's CPU table with n_keys resident keys and times the ops on it.  Every rate
  is the MEDIAN of 3 repeats of a fixed number of batches; the pool size is fixed."""
  try:
    _cpu_baseline_worker_body(conn, batch, n_keys, threads, kind, growth_leg, init_div)
  except BaseException as e:   # noqa: BLE001 — the parent reports it instead of a silent EOF
    try:
      conn.send(("error", "%s: %s" % (type(e).__name__, str(e)[:300])))
    except Exception:   # noqa: BLE001
      pass


def _cpu_baseline_worker_body(conn, batch, n_keys, threads, kind, growth_leg, init_div=1):
  import oracle
  dim = 64
  rng = np.random.default_rng(SEED)
  batches = [keys_of_ranks(zipf_bounded(rng, batch, n_keys)) for _ in range(8)]
  uniq = [np.unique(b) for b in batches]
  umean = float(np.mean([u.size for u in uniq]))
  vals = (rng.standard_normal((batch, dim)) * 0.01).astype(np.float32)
  zero = np.zeros(dim, np.float32)

  big = n_keys > 16_000_000
  chunk = 4_000_000 if big else 500_000
  rows = np.ascontiguousarray(np.broadcast_to(vals[:1], (min(chunk, n_keys), dim)))   # (materialised once: the fill must not time numpy copies)

  def fill(init_size):
    t0 = time.perf_counter()
    t = oracle.CpuTable(dim, np.float32, kind=kind, init_size=init_size, threads=threads)
    t_create = time.perf_counter() - t0
    conn.send(("phase", "created", round(t_create, 1)))
    t0 = time.perf_counter()
    for lo in range(1, n_keys + 1, chunk):
      k = keys_of_ranks(np.arange(lo, min(n_keys, lo + chunk - 1) + 1, dtype=np.int64))
      t.insert(k, rows[:k.size])
    conn.send(("phase", "filled", round(time.perf_counter() - t0, 1)))
    return t, t_create, time.perf_counter() - t0

  def rate(fn, units, iters=16, repeats=3):
    fn(0)
    out = []
    for r in range(repeats):
      t0 = time.perf_counter()
      for i in range(iters):
        fn(i)
      out.append(units * iters / (time.perf_counter() - t0))
    return sorted(out)[len(out) // 2]

  # init_div > 1: the table is created for n_keys / init_div keys and GROWS under the (multi-threaded) fill — the constructor of a
  # pre-sized libcuckoo table touches every bucket on ONE thread (a second per million keys), which a 256 M-key table cannot afford
  tab, create_s, fill_s = fill(max(8192, n_keys // init_div))
  ops = {
      # Find on the batch WITH its repeats (not what TFRA issues — it de-duplicates first): every occurrence of the hot id takes the
      # same bucket spinlock, 0.15 M ids/s on a 128-thread pool; three batches only (not at all on the 256 M-key rung: its time box)
      "find_with_repeats_ops_per_s": None if big else rate(lambda i: tab.find(batches[i % 8], zero), batch, iters=1),
      # what embedding_lookup issues: unique first (PY/dynamic_embedding_ops.py:99), Find on the distinct ids
      "find_unique_ids_ops_per_s": rate(lambda i: tab.find(uniq[i % 8], zero), umean),
      "insert_or_assign_ops_per_s": rate(lambda i: tab.insert(uniq[i % 8], vals[:uniq[i % 8].size]), umean),
      "insert_or_accum_ops_per_s": rate(lambda i: tab.accum(uniq[i % 8], vals[:uniq[i % 8].size], np.ones(uniq[i % 8].size, bool)), umean),
      # the step as the reference issues it: unique (timed apart, see dedup), Find(distinct ids), Insert(distinct ids, their rows);
      # counted in batch ids (pairs) per second
      "step_pairs_per_s": rate(lambda i: (tab.find(uniq[i % 8], zero), tab.insert(uniq[i % 8], vals[:uniq[i % 8].size])), batch),
      "prefill_keys_per_s_init_size_N": n_keys / fill_s,
  }
  del tab
  ded = []
  for r in range(3):
    t0 = time.perf_counter()
    for b in batches:
      np.unique(b, return_inverse=True)
    ded.append(len(batches) * batch / (time.perf_counter() - t0))
  grow_rate = None
  if growth_leg:   # growth included: the reference default init_size = 8192 (K/cuckoo_hashtable_op.cc:199-207), same fill
    tab, _, fill_s = fill(8192)
    grow_rate = n_keys / fill_s
    del tab
  conn.send(("done", {"ops": ops, "dedup_rate": sorted(ded)[1], "grow_rate": grow_rate, "create_s": create_s, "unique_per_batch": umean}))
  conn.close()


def cpu_baseline(batch):
  """The reference's CPU table — lib/cuckoo/cuckoohash_map.hh compiled in place (oracle/_ref) behind a restatement of
  TableWrapperOptimized + the LaunchTensors* launchers (K/cuckoo_hashtable_op.cc:39-182: static split of the keys over
  a persistent intra-op pool) — timed on this box's host cores: per op (find / insert_or_assign / insert_or_accum) and
  for the full lookup + write-back step, table pre-sized (init_size = N) and at the reference default (init_size = 8192,
  growth included).  dim 64 fp32 rows (256 B).  N = the largest rung of (256 M, 16 M, 4 M) resident keys whose table this
  box builds within its time box (SURVEY §8d asks for the largest N the RAM holds; the constructor of a pre-sized libcuckoo
  table touches every bucket on one thread — between 0.04 and 1 s per million keys depending on the box); each rung runs in a
  child process that is killed when it overruns.  Fixed pool size, every rate the median of 3 repeats."""
  import multiprocessing as mp
  import oracle
  kind = "reference" if oracle.available("reference") else "port"
  cores = os.cpu_count() or 1
  threads = min(cores, 128) if kind == "reference" else 1
  try:
    ram = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
  except (ValueError, OSError):
    ram = 0
  # (keys, init_size divisor, time box in s): 256 M keys (85 GB, pre-sized) on a box that has the cores and the RAM — on a fresh
  # round-4 box: constructor 9 s, 128-thread fill 33 s (7.7 M keys/s), the whole rung 47 s; the same rung started right after another
  # bench run on the same box did not get through its fill in 150 s (once; the host was still giving back the first run's 85 GB),
  # hence the box — 16 M as the fallback, 4 M as the last resort.  (Growing from N / 8 instead: 1 M keys/s, overran.)
  rungs = [r for r in ((256_000_000, 1, 100.0), (16_000_000, 1, 45.0), (4_000_000, 1, 90.0))
           if r[0] == 4_000_000 or (ram > 3 * r[0] * 330 and cores >= (64 if r[0] > 16_000_000 else 8))]
  if os.environ.get("TFRA_BENCH_CPU_KEYS"):
    rungs = [(int(os.environ["TFRA_BENCH_CPU_KEYS"]), int(os.environ.get("TFRA_BENCH_CPU_INIT_DIV", "8")), float(os.environ.get("TFRA_BENCH_CPU_BOX_S", "300")))]
  t_begin = time.perf_counter()
  got, tried = None, []
  ctx = mp.get_context("spawn")

  def run_rung(n_keys, init_div, box, growth_leg, note=None):
    """one table size in a child process, killed when it overruns its time box; -> the worker's result or None"""
    parent, child = ctx.Pipe(duplex=False)
    pr = ctx.Process(target=_cpu_baseline_worker, args=(child, batch, n_keys, threads, kind, growth_leg, init_div))
    pr.start()
    child.close()
    phases, t_rung, res, err = [], time.perf_counter(), None, None
    while res is None and err is None:   # the worker reports its phases; the last message is the result
      left = box - (time.perf_counter() - t_rung)
      if left <= 0 or not parent.poll(left):
        break
      try:
        msg = parent.recv()
      except EOFError:
        err = "the worker exited without a result"
        break
      if msg[0] == "done":
        res = msg[1]
      elif msg[0] == "error":
        err = msg[1]
      else:
        phases.append(list(msg[1:]))
    pr.join(timeout=1.0)
    if pr.is_alive():
      pr.kill()   # the exact process started above
      pr.join()
    rec = {"keys": n_keys, "finished": res is not None, "seconds": round(time.perf_counter() - t_begin, 1), "phases_s": phases}
    if err:
      rec["error"] = err
    if note:
      rec["legs"] = note
    tried.append(rec)
    return res

  # SURVEY §8d's two small-table legs — Find on a batch WITH its repeats, and the pre-fill at the reference default init_size = 8192
  # (growth included) — do not fit the big rung's time box: a 4 M-key rung of their own, FIRST (behind the big rung the host is still
  # giving back its 85 GB, and the small child died there once)
  small = None
  if rungs[0][0] > 16_000_000 and not os.environ.get("TFRA_BENCH_CPU_KEYS"):
    small = run_rung(4_000_000, 1, 60.0, True, "find_with_repeats, growth from init_size 8192")
  n_keys = init_div = None
  for n_keys, init_div, box in rungs:
    got = run_rung(n_keys, init_div, box, n_keys <= 16_000_000)
    if got is not None:
      break
  if got is None:
    return {"value": None, "unit": "lookup+insert pairs/s", "cores": threads, "kind": kind, "sample": "no rung finished: %s" % tried}
  if small is None and n_keys <= 16_000_000:
    small = got
  ops, dedup_rate = got["ops"], got["dedup_rate"]
  if small is not None and ops.get("find_with_repeats_ops_per_s") is None:
    ops["find_with_repeats_ops_per_s"] = small["ops"].get("find_with_repeats_ops_per_s")
  grow_rate = got["grow_rate"] or (small["grow_rate"] if small is not None else None)
  small_keys = 4_000_000 if (small is not None and small is not got) else n_keys
  # the whole CPU step = de-duplicate the batch (tf.unique in the reference, single-threaded; numpy's stands in) +
  # Find + Insert on the distinct ids
  step_incl_dedup = 1.0 / (1.0 / dedup_rate + 1.0 / ops["step_pairs_per_s"])
  return {
      "value": step_incl_dedup, "unit": "lookup+insert pairs/s", "cores": threads, "kind": kind,
      "table_ops_only_pairs_per_s": round(ops["step_pairs_per_s"]),
      "dedup_ids_per_s_numpy_unique_1_core": round(dedup_rate),
      "resident_keys": n_keys, "host_cores": cores, "host_ram_bytes": ram, "rungs_tried": tried,
      "sample": "per batch of %d Zipf-1.2 ids: unique + Find(distinct ids) + Insert(distinct ids) on a %d-key table (dim 64 fp32, "
                "init_size = N / %d, created in %.1f s), %d-thread pool (host has %d cores), median of 3 x 16 batches; %.0f s of CPU work"
                % (batch, n_keys, init_div, got["create_s"], threads, cores, time.perf_counter() - t_begin),
      "per_op": {k: (round(v) if v is not None else None) for k, v in ops.items()},
      "per_op_per_core": {k: (round(v / threads) if v is not None else None) for k, v in ops.items()},
      "prefill_keys_per_s_init_size_8192_growth_included": round(grow_rate) if grow_rate else None,
      "small_table_legs_keys": small_keys,
      "not_run_1e9_keys": "measured once (round 5, profiles/r05_cpu_baseline_1e9.json, TFRA_BENCH_CPU_KEYS=1000000000 TFRA_BENCH_CPU_INIT_DIV=1): "
                          "constructor 36 s on one thread + 128-thread fill 110 s + ops = 169 s of CPU work for 19.3 M pairs/s (table ops "
                          "alone 35 M) — six times the ~30 s the default run may spend on the CPU leg; the 256 M-key rung (85 GB, ~40 s) is the "
                          "largest that fits",
  }


# ------------------------------------------------------------------ helpers
class Timer:
  def __init__(self, torch):
    self.torch = torch
    self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

  def us(self, fn, reps=50, warm=5):
    """HIP events on torch's current stream (the stream every table call of this process is enqueued on)."""
    for i in range(warm):
      fn(i)
    self.e0.record()
    for i in range(reps):
      fn(warm + i)
    self.e1.record()
    self.torch.cuda.synchronize()
    return self.e0.elapsed_time(self.e1) * 1e3 / reps


def raw_calls(torch, dev):
  """Per-kernel timings go through the C ABI with pre-built ctypes arguments: a Python-level table call costs
  20-40 us of host time, more than most of these kernels, and would time the host instead."""
  import ctypes
  from tfra_amd import _capi
  lib = _capi.lib()
  st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
  P = lambda t: ctypes.c_void_p(t.data_ptr())

  class R:
    pass
  r = R()
  r.find = lambda h, ids, out, dflt: (lambda a=(h, ids.numel(), P(ids), P(out), None, P(dflt), 0, st): _capi.check(lib.tfra_table_find(*a)))
  r.upsert_planned = lambda h, plan, vals: (lambda a=(h, plan._h, P(vals), None, st): _capi.check(lib.tfra_table_upsert_planned(*a)))
  r.apply_planned = lambda h, p, plan, grads, dflt: (lambda a=(h, ctypes.byref(p), plan._h, P(grads), P(dflt), st): _capi.check(lib.tfra_table_apply_planned(*a)))
  r.plan_build = lambda plan, ids, dim: (lambda a=(plan._h, ids.numel(), P(ids), dim, st): _capi.check(lib.tfra_sparse_plan_build(*a)))
  r.apply_optimizer = lambda h, p, keys, grads, dflt: (lambda a=(h, ctypes.byref(p), keys.numel(), P(keys), P(grads), P(dflt), 0, None, st): _capi.check(lib.tfra_table_apply_optimizer(*a)))
  return r


WINDOWS = 5


def timed_windows(torch, dist, world, dev, K, step, first=0, windows=WINDOWS):
  """`windows` back-to-back windows of exactly K steps (step(first + w*K + i)), each bracketed by barrier +
  torch.cuda.synchronize() on both sides, the max over ranks taken per window.  Returns (per-window seconds, host seconds of
  the median window's enqueue loop).  The first window's first steps still run into cold caches / an idle clock: the median
  window is what the line reports, so that a driver run with --steps 20 and a builder run with --steps 200 agree."""
  secs, hosts = [], []
  for w in range(windows):
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
      step(first + w * K + i)
    host_s = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
      t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() != "gloo" else "cpu")
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      elapsed = float(t.item())
    secs.append(elapsed)
    hosts.append(host_s)
  order = sorted(range(windows), key=lambda i: secs[i])
  med = order[windows // 2]
  return secs, secs[med], hosts[med]


def note(msg):
  """progress to stderr (short lines: a driver may keep one merged tail) — where a run was when something went wrong"""
  if os.environ.get("RANK", "0") == "0":
    print("[bench] " + msg, file=sys.stderr, flush=True)


def timing_note(secs, K):
  return {"windows": len(secs), "steps_per_window": K, "reported": "median window",
          "ms_per_step_min": round(min(secs) / K * 1e3, 5), "ms_per_step_max": round(max(secs) / K * 1e3, 5),
          "ms_per_step_by_window": [round(x / K * 1e3, 5) for x in secs]}


def profile_summary():
  """the newest committed rocprofv3 summary (profiles/rNN_summary.json, made by scripts/run_profile_round.sh: separate --pmc FETCH_SIZE /
  WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes) — `roofline.traffic` is READ from it, not measured by this run:
  `roofline.traffic_source` names the file"""
  try:
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_summary.json"))
    if not cands:
      return None
    d = json.load(open(os.path.join(ROOT, "profiles", cands[-1])))
    d["_source"] = "profiles/" + cands[-1]
    return d
  except (OSError, ValueError):
    return None


def live_traffic(args):
  """`roofline.traffic` measured BY THIS RUN when rocprofv3 is here: two short child runs of the headline workload alone
  (`--config m1b --no-secondary --no-cpu-baseline --steps 10 --warmup 5`), one per counter as MI355X_MICROARCH.md prescribes
  (`rocprofv3 --pmc FETCH_SIZE --kernel-trace`, then `--pmc WRITE_SIZE --kernel-trace`: PMC passes on their own, never with API
  tracing), BEFORE this process touches the GPU (the table takes 273 of the 288 GB).  Per launch of the step kernel:
  (2 x FETCH_SIZE + WRITE_SIZE) KiB — gfx950 reports half of the wide coalesced reads.  Returns {"step_k": bytes, "source": ...} or None
  (no rocprofv3, a pass failed or ran into its limit: the line then quotes the committed profile summary and says so).
  TFRA_BENCH_LIVE_TRAFFIC=0 skips it."""
  import csv
  import glob
  import re
  import shutil
  import signal
  import subprocess
  import tempfile
  if os.environ.get("TFRA_BENCH_LIVE_TRAFFIC", "1") == "0":
    return None
  exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
  if not exe:
    return None
  tmp = tempfile.mkdtemp(prefix="tfra_live_", dir="/tmp")
  env = dict(os.environ)
  env.update({"TMPDIR": "/tmp", "TFRA_BENCH_LIVE_TRAFFIC": "0", "TFRA_BENCH_SKIP_ROUTED_LOCAL": "1"})
  per = {}
  try:
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
      d = os.path.join(tmp, c)
      cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "live", "--", sys.executable, os.path.abspath(__file__),
             "--config", "m1b", "--no-secondary", "--no-cpu-baseline", "--steps", "10", "--warmup", "5", "--slots", str(args.slots),
             "--batch", str(args.batch)]
      t0 = time.perf_counter()
      pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
      try:
        rc = pr.wait(timeout=float(os.environ.get("TFRA_BENCH_LIVE_TRAFFIC_LIMIT_S", "240")))
      except subprocess.TimeoutExpired:
        os.killpg(pr.pid, signal.SIGKILL)   # the process group started above (rocprofv3 and the bench under it), nothing else
        pr.wait()
        note("live traffic: the %s pass ran into its limit" % c)
        return None
      if rc != 0:
        note("live traffic: the %s pass ended with code %d" % (c, rc))
        return None
      tot, disp = 0.0, set()
      for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
          for row in csv.DictReader(fh):
            if row.get("Counter_Name") == c and re.search(r"\bstep_k_u\d", row.get("Kernel_Name", "")):
              tot += float(row["Counter_Value"])
              disp.add(row["Dispatch_Id"])
      if not disp:
        note("live traffic: no step launches in the %s pass" % c)
        return None
      per[c] = (tot / len(disp), len(disp))
      note("live traffic: %s %.1f KiB per step launch (%d launches, %.0f s)" % (c, per[c][0], per[c][1], time.perf_counter() - t0))
  except (OSError, ValueError, KeyError) as e:
    note("live traffic: %s" % e)
    return None
  finally:
    shutil.rmtree(tmp, ignore_errors=True)
  return {"step_k": int((2 * per["FETCH_SIZE"][0] + per["WRITE_SIZE"][0]) * 1024),
          "FETCH_SIZE_KiB_raw": round(per["FETCH_SIZE"][0], 1), "WRITE_SIZE_KiB": round(per["WRITE_SIZE"][0], 1), "launches": per["FETCH_SIZE"][1],
          "source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of THIS run (bench.live_traffic), (2 x FETCH + WRITE) KiB "
                    "per step launch, %d launches" % per["FETCH_SIZE"][1]}


def cpu_baseline_1e9():
  """the CPU leg at the metric's 10^9 resident keys, measured once on a GPU box's host (169 s of CPU work: it does not fit the default
  run, which stops at the 256 M-key rung) — quoted in the line with its source"""
  try:
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_cpu_baseline_1e9.json"))
    d = json.load(open(os.path.join(ROOT, "profiles", cands[-1])))
    return {"value_at_1e9_keys": d["value"], "table_ops_only_at_1e9_keys": d.get("table_ops_only_pairs_per_s"), "source_1e9": "profiles/" + cands[-1]}
  except (OSError, ValueError, IndexError, KeyError):
    return {}


def traffic_of(prof, workload, kname):
  try:
    return prof["workloads"][workload]["kernels"][kname]["hbm_bytes_per_launch_corrected"]
  except (TypeError, KeyError):
    return None


# ------------------------------------------------------------------ c3 / m1b: bounded table at 10^9 slots
def measure_growth(torch, de, dev, dim, dtype, slots):
  """configs[2] names "dynamic growth": a table of `slots` slots (a quarter of the benchmark's), half full, doubles IN PLACE
  (mapped address range, every bucket splits into its two children where it is: DESIGN §3).  Timed alone, then freed."""
  try:
    # (an Hkv table is created for init_capacity keys at load factor 0.5, on the lattice max_capacity / 2^j)
    t = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots // 2 - 1000, max_capacity=4 * slots, device=str(dev),
                        dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="bench_growth")
    cap0 = t._table.capacity() - 2
    n = int(cap0 * 0.45)
    chunk = 4_000_000
    vals = torch.zeros((chunk, dim), dtype=dtype, device=dev)
    for lo in range(1, n + 1, chunk):
      k = keys_of_ranks_torch(torch, torch.arange(lo, min(n, lo + chunk - 1) + 1, dtype=torch.int64, device=dev))
      t._table.upsert(k, vals[:k.numel()], unique_keys=True)
    size0 = int(t.size().item())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t._table.reserve(2 * cap0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = t._table.growth_stats()
    cap1 = t._table.capacity() - 2
    probe = keys_of_ranks_torch(torch, torch.arange(1, 1_000_001, dtype=torch.int64, device=dev))
    _, ex = t.lookup(probe, return_exists=True)
    ok = bool(ex.all()) and int(t.size().item()) == size0
    block = 256 + 15 * dim * (2 if dtype == torch.float16 else 4)
    res = {"slots_before": cap0, "slots_after": cap1, "live_keys": size0, "seconds": round(dt, 3), "in_place": st["in_place"] == 1,
           "bytes_before": cap0 // 15 * block, "bytes_after": cap1 // 15 * block, "all_keys_still_found": ok,
           "note": "tfra_table_reserve(2 x slots) on a half-full table: map the second half of the address range, split every bucket "
                   "into its two children where it is; peak memory = the new size (a copying growth would need old + new)"}
    del t, vals
    torch.cuda.empty_cache()
    return res
  except Exception as e:   # never lose the bench line over the side measurement
    return {"error": str(e)[:200]}


def last_occurrence_rows(torch, ids, values):
  """rows a lookup of `ids` must return after insert_or_assign(ids, values) with repeats: the row of each id's LAST position"""
  uk, inv = torch.unique(ids, return_inverse=True)
  lp = torch.zeros(uk.numel(), dtype=torch.long, device=ids.device)
  lp.scatter_reduce_(0, inv, torch.arange(ids.numel(), device=ids.device), reduce="amax", include_self=False)
  return values[lp][inv]


def run_bounded(args, torch, de, dev, cfg):
  """cfg 'm1b': dim 64 fp32, `--new-key-ratio` (default 0) never-seen ids;  'c3': dim 128 fp16, 50 % never-seen ids."""
  import ctypes
  from tfra_amd import _capi
  from tfra_amd.dynamic_embedding.table_ops import _ptr, _stream
  from tfra_amd.dynamic_embedding.device_ops import _workspace
  B, K, W = args.batch, args.steps, args.warmup
  dim, dtype = (128, torch.float16) if cfg == "c3" else (64, torch.float32)
  new_ratio = 0.5 if cfg == "c3" else (args.new_key_ratio if args.new_key_ratio is not None else 0.0)
  Rb = dim * (2 if dtype == torch.float16 else 4)
  want = args.slots
  growth = None
  if cfg == "c3":   # (the default invocation measures it FIRST, on a fresh device: right behind the release of another 273-GB table the
    growth = getattr(args, "_growth", None) or measure_growth(torch, de, dev, dim, dtype, want // 4)   # driver is still reclaiming memory, 6 s instead of 18 ms)
  table, failures = None, []
  for slots in [want] + [int(want * f) for f in (0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.25)]:
    try:
      table = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots, max_capacity=slots,
                              device=str(dev), dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="bench_%s" % cfg)
      break
    except Exception as e:  # allocation failure: next smaller size (logged in the result line)
      failures.append({"slots": slots, "error": str(e)[:160]})
  assert table is not None, failures
  gen = torch.Generator(device=dev).manual_seed(SEED)
  chunk = 4_000_000
  vals_fill = (torch.randn((chunk, dim), generator=gen, device=dev) * 0.01).to(dtype)
  t0 = time.perf_counter()
  n_res = slots
  # The pre-fill goes from the COLDEST ranks to the hottest: on an LRU table the order of the bulk load is the order of the scores.
  # (Rounds 1-3 loaded in rank order: the hottest ids of the Zipf stream were then the least recently used entries — the
  # first to be evicted, by the pre-fill itself and by the first steps, which no running table looks like.)
  for lo in range(((n_res - 1) // chunk) * chunk + 1, 0, -chunk):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(n_res, lo + chunk - 1) + 1, dtype=torch.int64, device=dev))
    table._table.upsert(k, vals_fill[:k.numel()], unique_keys=True)
  resident = int(table.size().item())
  t_fill = time.perf_counter() - t0
  del vals_fill
  tbl = table._table
  capacity = tbl.capacity()
  lib = _capi.lib()
  st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  idf = IdFactory(torch, dev, B, n_res, new_ratio, n_res + 1, SEED + 7)
  values = (torch.randn((B, dim), generator=gen, device=dev) * 0.01).to(dtype)
  nsteps = W + WINDOWS * K
  verified = {}
  tm = Timer(torch)
  # A stream with never-seen ids changes the table it meets after the bulk load (every step replaces its share of the least
  # recently used entries): the first driver measured used to pay for that transient — configs[2]'s shape ran the SAME step at 105 us
  # right after the pre-fill and at 45 us a few hundred steps later (scripts/mb_overlap.py --new-key-ratio 0.5).  All drivers are
  # timed on the table in its running state: 256 untimed steps of the workload first (plain calls).
  state_warm_steps = 256 if new_ratio > 0 else 0
  for lo in range(0, state_warm_steps, 32):
    wids = idf.keys(32)
    for i in range(32):
      tbl.find(wids[i])
      tbl.upsert_sparse(wids[i], values)
    del wids
  torch.cuda.synchronize()

  # ---- driver 1 (`value`): the overlapped step — ONE launch per step: lookup of batch i+1 (ids of batch i served from the rows
  # being written: store-to-load forwarding through batch i's plan), write-back of batch i, its left-over keys and the output
  # corrections (tail), and the plans of the next two batches (built over two launches, without atomics); D steps per host call
  # (tfra_table_steps_overlap), one stream, nothing waits on the host.  Lookup outputs: a ring of D buffers.
  uniq_ratio = None
  ovl = de.OverlapAssignStep(table)

  def time_overlapped(D):
    """W warm-up steps, then the windows: D steps per host call (tfra_table_steps_overlap with D pre-built argument blocks; D = 1 =
    one host call per step, what a trainer that gets the values of step i only after its lookup can issue)"""
    nonlocal uniq_ratio
    ids = idf.keys(nsteps + 3)
    if uniq_ratio is None:
      uniq_ratio = float(np.mean([torch.unique(ids[i]).numel() / B for i in range(4)]))
    outs = [torch.empty((B, dim), dtype=dtype, device=dev) for _ in range(max(D, 2))]
    ovl.prime(ids[0])
    for i in range(W):
      ovl.step(values, ids[i + 1], ids[i + 2])
    runs = [ovl.make_run([ids[W + c * D + q] for q in range(D)], [values] * D, [outs[(c * D + q) % len(outs)] for q in range(D)],
                         ids_after=ids[W + (c + 1) * D], values_before=values if (W + c) else None, ids_after2=ids[W + (c + 1) * D + 1])
            for c in range(WINDOWS * K // D)]

    def ovl_step(i):   # timed_windows calls once per step: every D-th call enqueues D steps (K is a multiple of D)
      if (i - W) % D == 0:
        runs[(i - W) // D]()

    r = timed_windows(torch, None, 1, dev, K, ovl_step, first=W)
    return r, ids, outs

  # the headline: ONE host call per step (D = 1)
  D = 1
  (secs, med, host_s), ids, outs = time_overlapped(1)
  # per-launch durations of the same driver: HIP events on the launching stream around each of the two launches of 24 more steps
  # (FRESH batches: re-used ones would find their never-seen ids resident by now — configs[2]'s launch timed on re-used ids showed 38 us
  # where the windows run at 70: no evictions left)
  kt_ids = idf.keys(24 + 3)
  ovl.time_kernels(24)
  for c in range(24):
    ovl.make_run([kt_ids[c]], [values], outs, ids_after=kt_ids[c + 1], values_before=values, ids_after2=kt_ids[c + 2])()
  ktimes = ovl.kernel_times()
  ovl.flush()
  ovl_stats = ovl.stats()
  size_after = int(table.size().item())
  last = kt_ids[24 - 1]
  got, ex = table.lookup(last, return_exists=True)
  verified["overlapped_step_last_batch"] = bool(ex.all()) and bool(torch.equal(got, last_occurrence_rows(torch, last, values)))
  # (the host learns that the table is dense from asynchronous size reads: the first warm-up steps may still run one op after the other)
  verified["overlapped_step_every_timed_step_overlapped"] = ovl_stats["sequential"] <= W and ovl_stats["overlapped"] >= WINDOWS * K + 24
  del got, ex, kt_ids
  # the same driver with 4 steps per host call (a bench-only mode: the values of steps i+1..i+3 would have to exist before their
  # lookups return) — reported beside the headline, never as `value`
  D4 = 4 if K % 4 == 0 else (2 if K % 2 == 0 else 1)
  (secs_d4, med_d4, host_d4), ids_d4, outs_d4 = time_overlapped(D4)
  ovl.flush()
  ovl_stats_d4 = ovl.stats()
  verified["overlapped_step_every_timed_step_overlapped"] = (verified["overlapped_step_every_timed_step_overlapped"] and
                                                             ovl_stats_d4["overlapped"] - ovl_stats["overlapped"] >= WINDOWS * K)
  del ovl, ids_d4, outs_d4
  note("%s: overlapped step timed (%.2f us/step)" % (cfg, med / K * 1e6))
  routed_local = None
  if cfg == "m1b" and os.environ.get("TFRA_BENCH_SKIP_ROUTED_LOCAL") != "1":   # (the profile runs skip it: its step launches would mix into the trace's step_k average)
    # what `--gpus N` runs per GPU (run_metric_sharded), at ONE rank on THIS table: the route driver with device copies where the alltoalls
    # would be — the route's own cost (route plan ahead; gather + copy + owner launch on the distinct ids + copy + gather per step)
    try:
      routed_local = routed_assign_measure(args, torch, None, de, dev, 1, 0, table, idf, values, "local", K, W, verified, "routed_local")
    except Exception as e:   # noqa: BLE001 — a side measurement must not lose the line
      routed_local = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    note("m1b: route driver at one rank timed")

  # ---- driver 1b: round 3's look-ahead driver (one C call per step, plan of batch i+1 on a second stream, host-ordered) --------
  # (FRESH batches here too: until round 6 this driver ran on the overlapped step's batches, whose never-seen ids were resident by then —
  # configs[2]'s write-back found every key and evicted nothing, and the line's c3 value was that of a stream without new keys)
  del ids
  ids = idf.keys(nsteps + 1)
  ps = de.PrefetchAssignStep(table).prime(ids[0])
  for i in range(W):
    ps.step(values, ids[i + 1])
  secs_pf, med_pf, host_pf = timed_windows(torch, None, 1, dev, K, lambda i: ps.step(values, ids[i + 1]), first=W)
  last = ids[nsteps - 1]
  got, ex = table.lookup(last, return_exists=True)
  verified["look_ahead_driver_last_batch"] = bool(ex.all()) and bool(torch.equal(got, last_occurrence_rows(torch, last, values)))
  del ps, ids, got, ex, outs

  note("%s: look-ahead driver timed" % cfg)
  # ---- driver 2: no look-ahead, two calls: Find, then the fused extra that de-duplicates inside the call --------------
  ids = idf.keys(nsteps)

  def plain(i):
    tbl.find(ids[i])
    tbl.upsert_sparse(ids[i], values)

  for i in range(W):
    plain(i)
  secs_plain, med_plain, _ = timed_windows(torch, None, 1, dev, K, plain, first=W)
  last = ids[nsteps - 1]
  got, ex = table.lookup(last, return_exists=True)
  verified["plain_call_last_batch"] = bool(ex.all()) and bool(torch.equal(got, last_occurrence_rows(torch, last, values)))
  del ids, got, ex

  # ---- driver 3: the reference's op surface as the TF shim issues it (tf_ops/mi355x_table_ops.h) -------------------
  #   Find op      -> tfra_table_find(B ids)
  #   tf.unique    -> tfra_unique + ONE host read of the count (the output shape of tf.unique is data dependent)
  #   Insert op    -> tfra_table_insert_or_assign(unique keys, rows, TFRA_FLAG_UNIQUE_KEYS)
  ids = idf.keys(nsteps)
  out_buf = torch.empty((B, dim), dtype=dtype, device=dev)
  dflt_row = tbl._default_value
  ubuf = torch.empty(B, dtype=torch.int64, device=dev)
  ibuf = torch.empty(B, dtype=torch.int32, device=dev)
  cbuf = torch.zeros((), dtype=torch.int64, device=dev)
  ws = _workspace(dev)
  finds = [(tbl._h, B, P(ids[i]), P(out_buf), None, P(dflt_row), 0, st) for i in range(nsteps)]
  uniqs = [(ws, B, P(ids[i]), P(ubuf), P(ibuf), P(cbuf), st) for i in range(nsteps)]
  last_u = [0]

  def op_surface_find_first(i):   # round 3's sequence: Find(B) -> unique + host read -> Insert(U)
    _capi.check(lib.tfra_table_find(*finds[i]))
    _capi.check(lib.tfra_unique(*uniqs[i]))
    u = int(cbuf.item())
    last_u[0] = u
    _capi.check(lib.tfra_table_insert_or_assign(tbl._h, u, P(ubuf), P(values), None, 1, st))

  for i in range(W):
    op_surface_find_first(i)
  secs_opf, med_opf, _ = timed_windows(torch, None, 1, dev, K, op_surface_find_first, first=W)
  u = last_u[0]
  got, ex = table.lookup(ubuf[:u], return_exists=True)
  verified["op_surface_find_first_last_batch"] = bool(ex.all()) and bool(torch.equal(got, values[:u]))

  # The reference's own order (embedding_lookup de-duplicates FIRST: PY/dynamic_embedding_ops.py:99-117,
  # PY/dynamic_embedding_variable.py:1377-1378):
  #   tf.unique(ids)          -> tfra_unique_unordered (two launches) + ONE host read of the count (tf.unique's output shape), here
  #                              through pinned memory the second launch writes to: a stream sync, no copy
  #   Find op (U ids)         -> tfra_table_find
  #   tf.gather(rows, idx)    -> tfra_gather_rows (B rows)
  #   Insert op (U ids)       -> tfra_table_insert_or_assign(TFRA_FLAG_UNIQUE_KEYS)
  cpin = torch.zeros(1, dtype=torch.int64).pin_memory()
  urows = torch.empty((B, dim), dtype=dtype, device=dev)
  uniqs2 = [(ws, B, P(ids[i]), P(ubuf), P(ibuf), ctypes.c_void_p(cpin.data_ptr()), st) for i in range(nsteps)]
  cur_stream = torch.cuda.current_stream(dev)
  row_bytes = dim * values.element_size()

  ev_u = torch.cuda.Event()
  cpin_p = ctypes.c_void_p(cpin.data_ptr())

  def op_surface(i):
    # Find / gather / Insert take the count from the device (tfra_table_find_n / tfra_table_insert_or_assign_n): they are enqueued
    # BEFORE the host reads it, and the read waits for the unique kernels only
    _capi.check(lib.tfra_unique_unordered(*uniqs2[i]))
    ev_u.record(cur_stream)
    _capi.check(lib.tfra_table_find_n(tbl._h, B, cpin_p, P(ubuf), P(urows), None, P(dflt_row), 0, st))
    _capi.check(lib.tfra_gather_rows(B, row_bytes, P(urows), P(ibuf), P(out_buf), st))
    _capi.check(lib.tfra_table_insert_or_assign_n(tbl._h, B, cpin_p, P(ubuf), P(values), None, st))
    ev_u.synchronize()
    last_u[0] = int(cpin[0])      # tf.unique's output shape

  def op_surface_sync(i):   # the same with the host read IN FRONT of Find (round 4's first form)
    _capi.check(lib.tfra_unique_unordered(*uniqs2[i]))
    cur_stream.synchronize()
    u = int(cpin[0])
    last_u[0] = u
    _capi.check(lib.tfra_table_find(tbl._h, u, P(ubuf), P(urows), None, P(dflt_row), 0, st))
    _capi.check(lib.tfra_gather_rows(B, row_bytes, P(urows), P(ibuf), P(out_buf), st))
    _capi.check(lib.tfra_table_insert_or_assign(tbl._h, u, P(ubuf), P(values), None, 1, st))

  def op_surface_fused(i):
    # what the registered fused TF ops issue (tf_ops/fused_ops_rocm.cc): TFRA>HkvHashTableEmbeddingLookup = Find of all B ids + the
    # unique ids / inverse index for the backward pass in ONE launch (tfra_table_find_unique), TFRA>HkvHashTableInsertN = Insert of the unique keys, count on the device —
    # no host read anywhere (every output has the upper-bound shape [B])
    _capi.check(lib.tfra_table_find_unique(*fus[i]))
    _capi.check(lib.tfra_table_insert_or_assign_n(tbl._h, B, cpin_p, P(ubuf), P(values), None, st))

  fus = [(tbl._h, ws, B, P(ids[i]), P(out_buf), None, P(dflt_row), 0, P(ubuf), P(ibuf), cpin_p, st) for i in range(nsteps)]
  # (every variant on batches of its own — new never-seen ids where the workload has them: the argument blocks point into `ids`, which is
  # refilled in place)
  ids.copy_(idf.keys(nsteps))
  for i in range(W):
    op_surface_fused(i)
  secs_opfu, med_opfu, _ = timed_windows(torch, None, 1, dev, K, op_surface_fused, first=W)
  torch.cuda.synchronize()
  u = int(cpin[0])
  got, ex = table.lookup(ubuf[:u], return_exists=True)
  verified["op_surface_fused_ops_last_batch"] = bool(ex.all()) and bool(torch.equal(got, values[:u]))
  ids.copy_(idf.keys(nsteps))
  for i in range(W):
    op_surface(i)
  secs_ops, med_ops, _ = timed_windows(torch, None, 1, dev, K, op_surface, first=W)
  torch.cuda.synchronize()
  ids.copy_(idf.keys(nsteps))
  for i in range(W):
    op_surface_sync(i)
  secs_ops_sync, med_ops_sync, _ = timed_windows(torch, None, 1, dev, K, op_surface_sync, first=W)
  u = last_u[0]
  got, ex = table.lookup(ubuf[:u], return_exists=True)
  verified["op_surface_last_batch"] = bool(ex.all()) and bool(torch.equal(got, values[:u]))
  # (the gather really returned the rows of the batch's ids: the lookup of the LAST batch, taken before its insert, against a plain find)
  chk = idf.keys(1)[0]
  _capi.check(lib.tfra_unique_unordered(ws, B, P(chk), P(ubuf), P(ibuf), ctypes.c_void_p(cpin.data_ptr()), st))
  cur_stream.synchronize()
  uu = int(cpin[0])
  _capi.check(lib.tfra_table_find(tbl._h, uu, P(ubuf), P(urows), None, P(dflt_row), 0, st))
  _capi.check(lib.tfra_gather_rows(B, row_bytes, P(urows), P(ibuf), P(out_buf), st))
  verified["op_surface_unique_find_gather_equals_find"] = bool(torch.equal(out_buf, table.lookup(chk)))

  # bp_v2 (`Accum` op, K/hkv_hashtable_op_gpu.cu.cc:292-335): Find(B) -> insert_or_accum of the batch's unique keys (prepared
  # beforehand like the table-ops-only line below; every key exists: row += delta)
  uqa = [torch.unique(ids[i]) for i in range(nsteps)]
  ex_true = torch.ones(B, dtype=torch.bool, device=dev)

  def accum_step(i):
    _capi.check(lib.tfra_table_find(*finds[i]))
    _capi.check(lib.tfra_table_accum_or_assign(tbl._h, uqa[i].numel(), P(uqa[i]), P(values), P(ex_true), None, 1, st))

  before = table.lookup(uqa[nsteps - 1])
  for i in range(W):
    accum_step(i)
  secs_acc, med_acc, _ = timed_windows(torch, None, 1, dev, K, accum_step, first=W)
  got, ex = table.lookup(uqa[nsteps - 1], return_exists=True)
  nu = uqa[nsteps - 1].numel()
  # (the last batch's keys: present ones got exactly one more delta in the last step — rows other batches also touched moved further)
  verified["accum_last_batch_rows_moved"] = bool(ex.any()) and not bool(torch.equal(got, before))
  del before, uqa
  # the two table ops alone, the unique keys prepared beforehand
  uq = [torch.unique(ids[i]) for i in range(nsteps)]
  ins = [(tbl._h, uq[i].numel(), P(uq[i]), P(values), None, 1, st) for i in range(nsteps)]

  def table_ops_only(i):
    _capi.check(lib.tfra_table_find(*finds[i]))
    _capi.check(lib.tfra_table_insert_or_assign(*ins[i]))

  for i in range(W):
    table_ops_only(i)
  secs_tops, med_tops, _ = timed_windows(torch, None, 1, dev, K, table_ops_only, first=W)
  got, ex = table.lookup(uq[nsteps - 1], return_exists=True)
  verified["op_surface_table_ops_only_last_batch"] = bool(ex.all()) and bool(torch.equal(got, values[:uq[nsteps - 1].numel()]))
  del got, ex

  note("%s: op-surface variants timed" % cfg)
  # ---- per-kernel timings (HIP events on the launching stream), fresh batches each launch ----------------------
  rc = raw_calls(torch, dev)
  NP = 24                                     # write-back launches timed, every one on its own batch (6 were too few: +-10 %)
  kids = idf.keys(6 + 2 * NP)
  fk = [rc.find(tbl._h, kids[j], out_buf, dflt_row) for j in range(6)]
  find_us = tm.us(lambda i: fk[i % 6](), reps=24, warm=3)
  plans = [de.table_ops.SparsePlan(dev, 0) for _ in range(NP)]
  for j, pl in enumerate(plans):
    pl.build(kids[6 + j], sync=False)
  torch.cuda.synchronize()
  counts = plans[0].read()[0]
  U = counts["many"] + counts["few"]
  ups = [rc.upsert_planned(tbl._h, plans[j], values) for j in range(NP)]
  upsert_us = tm.us(lambda i: ups[i](), reps=NP - 4, warm=4)   # every launch writes its own batch
  builds = [rc.plan_build(plans[j], kids[6 + j], 0) for j in range(8)]
  plan_us = tm.us(lambda i: builds[i % 8](), reps=24, warm=3)
  uqk = [torch.unique(kids[6 + NP + j]) for j in range(NP)]
  insk = [(lambda a=(tbl._h, uqk[j].numel(), P(uqk[j]), P(values), None, 1, st): _capi.check(lib.tfra_table_insert_or_assign(*a))) for j in range(NP)]
  insert_unique_us = tm.us(lambda i: insk[i](), reps=NP - 4, warm=4)
  unique_us = tm.us(lambda i: _capi.check(lib.tfra_unique(*uniqs[i % nsteps])), reps=24, warm=3)
  export = None
  if cfg == "c3":
    # export: one window of 16 Mi slots per launch, a different window each time; a full sweep = capacity / window launches
    win = min(16 << 20, capacity)
    kbuf = torch.empty(win, dtype=torch.int64, device=dev)
    vbuf = torch.empty((win, dim), dtype=dtype, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    nwin = max(1, capacity // win)

    def export_window(i):
      cnt.zero_()
      _capi.call("tfra_table_export_batch", tbl._h, win, (i % nwin) * win, _ptr(cnt), _ptr(kbuf), _ptr(vbuf), None, _stream(dev))

    export_us = tm.us(export_window, reps=6, warm=1)
    live = int(cnt.item())
    export_bytes = (win // 15 + 1) * (4096 if Rb == 256 else 256 + 15 * Rb) + live * (8 + Rb)
    sweep_s = export_us * 1e-6 * (capacity / win)
    del kbuf, vbuf
    export = {"window_slots": win, "live_keys_in_last_window": live, "avg_launch_us": export_us,
              "achieved_GBps": export_bytes / export_us / 1e3, "full_sweep_s": sweep_s,
              "pairs_per_s_with_a_full_export_sweep_every_1000_steps": B * 1000 / (1000 * med / K + sweep_s),
              "note": "export_batch(n, offset) windows over the slot range (K/lookup_impl/lookup_table_op_hkv.h:548-594); BASELINE's "
                      "'export every 1000 steps' is NOT inside the timed region (%d steps): the sweep is timed alone and folded in "
                      "arithmetically" % (WINDOWS * K)}
  # ---- untimed verification of the table itself (SURVEY §8c properties that do not depend on the size) ---------------
  tbl.check_errors()
  census = {k: int(v) for k, v in tbl.slot_census().items()}
  size_end = int(table.size().item())
  verified.update({"check_errors_clean": True, "size_le_capacity": size_end <= capacity, "no_locked_slot": census["locked"] == 0,
                   "size_matches_live_slots": abs(size_end - census["live"]) <= 2, "size_at_end": size_end})
  bad = [k for k, v in verified.items() if v is False]
  assert not bad, "bench verification failed: %s (overlapped step: %s)" % (bad, ovl_stats)

  # `value` = ONE driver per workload, chosen by RULE (de.assign_step_driver_for: the stream's share of never-seen ids), not by
  # result: the overlapped step for streams that mostly revisit resident keys (the metric's), the look-ahead driver when more than
  # a tenth of a batch are never-seen ids (configs[2]: the write-back with its evictions is the long pole and runs faster as
  # kernels of its own).  The other driver's number is reported beside it.
  best_is_overlapped = de.assign_step_driver_for(new_ratio) == "overlapped_step"
  med_best = med if best_is_overlapped else med_pf
  ms = med_best / K * 1e3
  value = B * K / med_best
  lookup_bytes = B * (8 + 2 * Rb)                      # key + row read + row written out
  upsert_bytes = U * (8 + Rb + Rb + 8)                 # per unique key: key, value row read, row written, key stored
  step_bytes = step_bytes_ = lookup_bytes + upsert_bytes
  prof = profile_summary()
  live = getattr(args, "_live_traffic", None) if cfg == "m1b" else None
  kernels = {
      "step_k (the step's ONE launch: lookup of batch i+1 with store-to-load forwarding + ownership write-back of batch i + its tail + plan builders)": {
          "avg_launch_us": ktimes["step_kernel_us"], "algorithmic_bytes_per_launch": step_bytes_, "unique_keys": U,
          "achieved_GBps": step_bytes_ / ktimes["step_kernel_us"] / 1e3, "frac": step_bytes_ / ktimes["step_kernel_us"] / 1e3 / HBM_PEAK_GBS,
          "traffic": (live or {}).get("step_k") or traffic_of(prof, cfg, "step_k"), "launches_timed": ktimes["steps"]},
      "find_kernel<16,4,WT,PF1> (lookup; both home-bucket lines in flight)": {
          "avg_launch_us": find_us, "algorithmic_bytes_per_launch": lookup_bytes,
          "achieved_GBps": lookup_bytes / find_us / 1e3, "frac": lookup_bytes / find_us / 1e3 / HBM_PEAK_GBS,
          "traffic": traffic_of(prof, cfg, "find_kernel")},
      "upsert_own_kernel<16,SIMPLE,SET> + upsert_rest_kernel<16,SET> (write-back of a plan's distinct keys: single pass with bucket ownership, then the few keys that lost a claim)": {
          "avg_launch_us": upsert_us, "algorithmic_bytes_per_launch": upsert_bytes, "unique_keys": U,
          "achieved_GBps": upsert_bytes / upsert_us / 1e3, "frac": upsert_bytes / upsert_us / 1e3 / HBM_PEAK_GBS,
          "traffic": (traffic_of(prof, cfg, "upsert_own_kernel[set]") or 0) + (traffic_of(prof, cfg, "upsert_rest_kernel[set]") or 0) or None},
      "upsert_own_kernel<16,SIMPLE,DIRECT> + upsert_rest_kernel<16,DIRECT> (tfra_table_insert_or_assign of unique keys: the reference's Insert op)": {
          "avg_launch_us": insert_unique_us, "algorithmic_bytes_per_launch": upsert_bytes, "unique_keys": U,
          "achieved_GBps": upsert_bytes / insert_unique_us / 1e3, "frac": upsert_bytes / insert_unique_us / 1e3 / HBM_PEAK_GBS,
          "traffic": (traffic_of(prof, cfg, "upsert_own_kernel[direct]") or 0) + (traffic_of(prof, cfg, "upsert_rest_kernel[direct]") or 0) or None},
  }
  on_step = list(kernels)[:1] if best_is_overlapped else list(kernels)[1:3]   # the launches of the `value` step
  dom = max(on_step, key=lambda k: kernels[k]["avg_launch_us"])
  cfg_name = "2" if cfg == "c3" else "metric (dim 64 fp32, 1 B keys, Zipf-1.2)"
  res = {
      "metric": "embedding lookup+insert pairs/s (%s, %d-slot bounded table, %d %% never-seen ids per batch: lookup + "
                "insert_or_assign with score-based eviction)" % ("dim=128 fp16" if cfg == "c3" else "dim=64 fp32", capacity,
                                                                 round(100 * new_ratio)),
      "value": value, "unit": "pairs/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": ms,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "f32",
      "data": "synthetic",
      "value_overlapped_step": B * K / med, "ms_per_step_overlapped_step": med / K * 1e3,
      "value_overlapped_step_4_steps_per_host_call": B * K / med_d4, "ms_per_step_overlapped_step_4_steps_per_host_call": med_d4 / K * 1e3,
      "value_look_ahead_driver": B * K / med_pf, "ms_per_step_look_ahead_driver": med_pf / K * 1e3,
      "driver": "overlapped_step" if best_is_overlapped else "look_ahead",
      "driver_rule": "de.assign_step_driver_for(new_key_ratio): overlapped_step when <= 10 %% of a batch are never-seen ids, else look_ahead "
                     "(new_key_ratio here: %.2f)" % new_ratio,
      "routed_local": None if routed_local is None else (routed_local if "error" in routed_local else {
          "value": B * K / routed_local["med"], "ms_per_step": routed_local["med"] / K * 1e3, "owner_launch_us": routed_local["owner_launch_us"],
          "host_enqueue_ms_per_step": round(1e3 * routed_local["host_s"] / K, 4), "served_ids_per_step": routed_local["served_ids"],
          "route_stats": routed_local["stats"], "timing": timing_note(routed_local["secs"], K)}),
      "value_plain_call": B * K / med_plain, "ms_per_step_plain_call": med_plain / K * 1e3,
      "value_op_surface": B * K / med_ops, "ms_per_step_op_surface": med_ops / K * 1e3,
      "value_op_surface_fused_ops": B * K / med_opfu, "ms_per_step_op_surface_fused_ops": med_opfu / K * 1e3,
      "value_op_surface_host_read_first": B * K / med_ops_sync, "ms_per_step_op_surface_host_read_first": med_ops_sync / K * 1e3,
      "value_op_surface_find_first": B * K / med_opf, "ms_per_step_op_surface_find_first": med_opf / K * 1e3,
      "value_accum": B * K / med_acc, "ms_per_step_accum": med_acc / K * 1e3,
      "value_op_surface_table_ops_only": B * K / med_tops, "ms_per_step_op_surface_table_ops_only": med_tops / K * 1e3,
      "config": {
          "workload": "BASELINE configs[%s]: bounded Hkv (LRU) table, %d slots (%.1f GB in HBM: key line + score line + 15 rows per "
                      "4 KiB bucket block), pre-filled with %d unique keys -> %d resident, %s rows, batch=%d = %d %% Zipf-1.2 over "
                      "the %d ranks + %d %% never-seen ranks; step = lookup(B) + insert_or_assign(B, last occurrence "
                      "wins) with eviction%s" % (cfg_name, capacity, capacity / 15 * (256 + 15 * Rb) / 1e9, n_res, resident,
                                                  "dim=128 fp16" if cfg == "c3" else "dim=64 fp32", B, round(100 * (1 - new_ratio)), n_res,
                                                  round(100 * new_ratio),
                                                  "; the export sweep of configs[2] ('every 1000 steps') is outside the timed region, see config.export" if cfg == "c3" else ""),
          "slots": capacity, "requested_slots": want, "alloc_failures": failures, "growth_in_place": growth,
          "resident_after_prefill": resident,
          "resident_after_timed_steps": size_after, "new_key_ratio": new_ratio, "global_batch": B,
          "unique_ratio": round(uniq_ratio, 4), "unique_keys_per_batch": U, "prefill_s": round(t_fill, 2),
          "state_warm_steps_before_any_timing": state_warm_steps,
          "table_ops_per_s": (B + U) * K / med_best,
          "table_ops_per_s_counts": "B lookups + U row writes (the distinct keys of the batch) per step",
          "host_enqueue_ms_per_step": round(1e3 * (host_s if best_is_overlapped else host_pf) / K, 4), "steps_per_host_call": 1,
          "host_enqueue_ms_per_step_overlapped_step": round(1e3 * host_s / K, 4),
          "host_enqueue_ms_per_step_overlapped_step_4_steps_per_host_call": round(1e3 * host_d4 / K, 4),
          "host_enqueue_ms_per_step_look_ahead_driver": round(1e3 * host_pf / K, 4),
          "overlapped_step_stats": ovl_stats,
          "timing": {"value_overlapped_step": timing_note(secs, K), "value_overlapped_step_4_steps_per_host_call": timing_note(secs_d4, K), "value_look_ahead_driver": timing_note(secs_pf, K), "value_plain_call": timing_note(secs_plain, K),
                     "value_op_surface": timing_note(secs_ops, K), "value_op_surface_fused_ops": timing_note(secs_opfu, K), "value_op_surface_find_first": timing_note(secs_opf, K), "value_op_surface_host_read_first": timing_note(secs_ops_sync, K),
                     "value_accum": timing_note(secs_acc, K), "value_op_surface_table_ops_only": timing_note(secs_tops, K)},
          "verified": verified,
          "drivers": {
              "value": "= value_overlapped_step" if best_is_overlapped else "= value_look_ahead_driver",
              "value_overlapped_step": "tfra_table_steps_overlap (csrc/tfra_step_impl.h): ONE launch per step on one stream = lookup of batch i+1 (ids of "
                       "batch i served from the rows being written: store-to-load forwarding through batch i's plan) + ownership "
                       "write-back of batch i + its left-over keys and the output corrections (tail blocks of the same launch) + the "
                       "plans of batches i+2 / i+3 (built over two launches, no atomics); results identical to lookup; insert; lookup; "
                       "insert ... (needs the ids two batches ahead); ONE host call per step (value_overlapped_step_4_steps_per_host_call: %d "
                       "steps per call of tfra_table_steps_overlap — bench-only, a trainer has the values of step i only after its lookup)" % D4,
              "value_look_ahead_driver": "round 3's driver, tfra_table_step_prefetch_assign: ONE C call per step = lookup + insert_or_assign "
                                         "of batch i on the main stream, plan of batch i+1 on a second stream, streams ordered by the host",
              "value_plain_call": "tfra_table_find then tfra_table_upsert_sparse, no look-ahead; upsert_sparse is a fused extra (plan "
                                  "built inside the call, repeats resolved on the device), not an op of the reference's surface",
              "value_op_surface": "the reference's own op order (embedding_lookup de-duplicates first, PY/dynamic_embedding_ops.py:99-117): "
                                  "tfra_unique_unordered (B ids; count into pinned memory) -> tfra_table_find_n -> tfra_gather_rows (B "
                                  "rows) -> tfra_table_insert_or_assign_n (unique keys): Find and Insert read the count on the device and "
                                  "are enqueued BEFORE the one host read of it (tf.unique's output shape), which waits for the unique "
                                  "kernels only",
              "value_op_surface_fused_ops": "the registered fused TF ops' call sequence (tf_ops/fused_ops_rocm.cc): tfra_table_find_unique = ONE launch with the find of "
                                            "the B ids next to their de-duplication (unique ids, inverse index, count on the device) = TFRA>HkvHashTableEmbeddingLookup, "
                                            "tfra_table_insert_or_assign_n = TFRA>HkvHashTableInsertN; no host read",
              "value_op_surface_host_read_first": "the same ops with the host read in front of Find (tfra_table_find / "
                                                  "tfra_table_insert_or_assign called with the count): the calls of tf_ops/mi355x_table_ops.h",
              "value_op_surface_find_first": "round 3's sequence: tfra_table_find (B ids) -> tfra_unique (ordered) + one blocking host read "
                                             "-> tfra_table_insert_or_assign(U unique keys)",
              "value_accum": "bp_v2: tfra_table_find (B ids) -> tfra_table_accum_or_assign(U unique keys prepared beforehand, exists = "
                             "true: row += delta, TFRA_FLAG_UNIQUE_KEYS) on the single-pass ownership kernels",
              "value_op_surface_table_ops_only": "tfra_table_find (B ids) -> tfra_table_insert_or_assign(unique keys prepared beforehand)"},
          "plan_build_us_alone": plan_us, "tfra_unique_us_alone": unique_us,
          "export": export,
      },
      "roofline": {
          "bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": kernels[dom]["frac"], "traffic": kernels[dom]["traffic"],
          "traffic_source": ((live["source"] if (live and dom.startswith("step_k")) else (prof or {}).get("_source"))
                             if kernels[dom]["traffic"] is not None else None),
          "traffic_profile_summary": traffic_of(prof, cfg, "step_k") if (live and dom.startswith("step_k")) else None,
          "avg_launch_us_note": "HIP events around the launch on its stream: one dispatch (~1.5 us) more than the kernel itself (rocprofv3's per-kernel "
                                "average of the same build: profiles/*_kernel_stats.csv)",
          "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes_per_launch"], "avg_launch_us": kernels[dom]["avg_launch_us"],
          "step_frac": step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
          "step_frac_look_ahead_driver": step_bytes / (med_pf / K) / 1e9 / HBM_PEAK_GBS,
          "step_frac_plain_call": step_bytes / (med_plain / K) / 1e9 / HBM_PEAK_GBS,
          "step_frac_op_surface": step_bytes / (med_ops / K) / 1e9 / HBM_PEAK_GBS,
          "step_frac_accum": (lookup_bytes + U * (9 + 3 * Rb)) / (med_acc / K) / 1e9 / HBM_PEAK_GBS,
          "step_algorithmic_bytes": step_bytes,
          "step_bytes_definition": "B*(8+2*Rb) for the lookup + U*(16+2*Rb) for the write-back of the U unique keys (SURVEY §8d)",
          "by_survey_pair_count": B * 1048 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
          "kernels": kernels,
          "timing": "HIP events on the launching stream around 20-24 launches, a different batch each (step_k: recorded by the step "
                    "driver around its launch); the fused launch moves ALL of the step's algorithmic bytes, so its fraction is the step's",
      },
  }
  del plans, table, tbl, uq, ins, finds
  import gc
  gc.collect()
  torch.cuda.empty_cache()
  return res


# ------------------------------------------------------------------ the metric's workload, hash-sharded over the ranks (N > 1; N = 1: --config m1s)
def gather_served_counts(torch, dist, world, rank, ids):
  """(ids this rank serves for one step, distinct ones among them) when every rank routes the distinct ids of a batch like `ids`:
  untimed, for the owner launch's algorithmic bytes"""
  uq = torch.unique(ids)
  if world == 1:
    return int(uq.numel()), int(uq.numel())
  # the batches differ per rank: every rank's share for THIS owner, through the host (small: a few thousand ids per rank)
  sends = [uq[((uq & 0x7FFFFFFF) % world) == r].cpu().numpy() for r in range(world)]
  gathered = [None] * world
  dist.all_gather_object(gathered, sends)
  got = np.concatenate([gathered[src][rank] for src in range(world)])
  return int(got.size), int(np.unique(got).size)


def routed_assign_measure(args, torch, dist, de, dev, world, rank, table, idf, values, transport, K, W, verified, tag):
  """W warm-up steps + the timed windows of RoutedAssignStep (tfra_assign_route_*: csrc/tfra_aroute.hip) on `table` = this rank's shard;
  ONE step call + ONE feed call per step through the C ABI with pre-built arguments, five batches fed ahead; then 24 steps with HIP events
  around the owner's step launch, a flush, and the untimed check of the last batch against every rank's writes."""
  import ctypes
  from tfra_amd import _capi
  from tfra_amd.dynamic_embedding.distributed import RoutedAssignStep
  lib = _capi.lib()
  B = args.batch
  dim, dtype = values.shape[1], values.dtype
  AHEAD = int(os.environ.get("TFRA_ROUTE_AHEAD", "5"))
  Wr = max(W, 8)   # the routed pipeline runs five batches ahead on its own streams: the first steps fill it
  NT = 24
  rs = RoutedAssignStep(table, transport=transport, max_batch=B)
  assert not rs.identity
  st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  nb = Wr + WINDOWS * K + NT + 1
  ids = idf.keys(nb + AHEAD + 1)
  outs = [torch.empty((B, dim), dtype=dtype, device=dev) for _ in range(2)]
  dflt = rs.default
  torch.cuda.synchronize()   # the ids are complete: the route's own stream reads them without an event (ids_ready = 1)
  feeds = [(rs._h, B, P(ids[i]), 1, st) for i in range(nb + AHEAD + 1)]
  step_first = (rs._h, P(outs[0]), P(dflt), None, st)
  steps = [(rs._h, P(outs[q]), P(dflt), P(values), st) for q in range(2)]

  def call(fn, a):
    rc = fn(*a)
    if rc:
      if rs._staged is not None and rs._staged.error is not None:
        raise rs._staged.error
      _capi.check(rc)

  for j in range(AHEAD + 1):
    call(lib.tfra_assign_route_feed, feeds[j])

  def step(i):
    call(lib.tfra_assign_route_step, step_first if i == 0 else steps[i & 1])
    call(lib.tfra_assign_route_feed, feeds[i + AHEAD + 1])

  for i in range(Wr):
    step(i)
  secs, med, host_s = timed_windows(torch, dist, world, dev, K, step, first=Wr)
  _capi.check(lib.tfra_assign_route_time_kernels(rs._h, NT))
  first_t = Wr + WINDOWS * K
  for i in range(first_t, first_t + NT):
    step(i)
  us, nst = ctypes.c_double(), ctypes.c_size_t()
  _capi.check(lib.tfra_assign_route_kernel_times(rs._h, ctypes.byref(us), ctypes.byref(nst)))
  # the batch looked up last (first_t + NT - 1) is pending: look its ids up ONCE MORE — the step writes it back first
  last = ids[first_t + NT - 1]
  for k in range(AHEAD + 1):   # drain what is fed ahead (each is written back with `values`), then the last batch again
    call(lib.tfra_assign_route_step, steps[k & 1])
  call(lib.tfra_assign_route_feed, (rs._h, B, P(last), 1, st))
  chk = torch.empty((B, dim), dtype=dtype, device=dev)
  call(lib.tfra_assign_route_step, (rs._h, P(chk), P(dflt), P(values), st))
  call(lib.tfra_assign_route_flush, (rs._h, P(values), st))
  torch.cuda.synchronize()
  # expected: the row of a key of `last` = its LAST write over rank 0's, rank 1's, ... batches (every rank wrote ITS `values` at the key's
  # last position: the highest rank holding the key wins) — for the keys none of the drained batches (which came later and wrote too) holds
  later = torch.unique(torch.cat([ids[first_t + NT + k] for k in range(AHEAD + 1)]))
  if world > 1:
    on_dev = dist.get_backend() == "nccl"
    mv = lambda t: t if on_dev else t.cpu()
    g_last = [torch.empty_like(mv(last)) for _ in range(world)]
    dist.all_gather(g_last, mv(last))
    g_vals = [torch.empty_like(mv(values)) for _ in range(world)]
    dist.all_gather(g_vals, mv(values))
    later_all = [None] * world
    dist.all_gather_object(later_all, later.cpu().numpy())
    later = torch.from_numpy(np.unique(np.concatenate(later_all))).to(dev)
    all_ids = torch.cat([t.to(dev) for t in g_last]); all_vals = torch.cat([t.to(dev) for t in g_vals])
    want = last_occurrence_rows(torch, all_ids, all_vals)[rank * B:(rank + 1) * B]
    del g_last, g_vals, all_ids, all_vals
  else:
    want = last_occurrence_rows(torch, last, values)
  keep = ~torch.isin(last, later)
  diff = keep & (chk != want).any(dim=1)
  # (a bounded table at capacity: a key of `last` may have been EVICTED by one of the drained batches' new keys — it then reads as the
  # default row; anything else is an error)
  evicted_only = bool((chk[diff] == dflt.to(chk.dtype)).all()) if bool(diff.any()) else True
  verified["%s_last_batch" % tag] = bool(keep.any()) and evicted_only and int(diff.sum()) <= max(8, int(keep.sum()) // 500)
  stats = rs.stats()
  verified["%s_every_owner_step_overlapped" % tag] = stats["owner_sequential"] <= Wr
  nr, nd = gather_served_counts(torch, dist, world, rank, last)
  rccl = rs.rccl_ranks
  rs.close()
  return {"secs": secs, "med": med, "host_s": host_s, "owner_launch_us": us.value, "launches_timed": nst.value, "stats": stats,
          "served_ids": nr, "served_distinct": nd, "rccl_ranks": rccl, "ahead": AHEAD, "warm": Wr}


def run_metric_sharded(args, torch, dist, de, dev, world, rank):
  """The metric's step — lookup(B) + insert_or_assign(B, last occurrence wins) on a bounded LRU table at capacity, dim 64 fp32,
  Zipf-1.2 — with the table HASH-SHARDED over the ranks (BASELINE configs[3]'s sharding: owner = default_partition_fn, 5*10^8 slots
  per GPU by default), per-GPU batch B drawn from the GLOBAL Zipf over all world*slots ranks, ids / rows / values routed
  (RoutedAssignStep), the owner running the overlapped step on what it receives.  At world 1 (`--config m1s`) the same driver with
  device copies where the alltoalls would be: what the route itself costs."""
  B, K, W = args.batch, args.steps, args.warmup
  dim, dtype, Rb = 64, torch.float32, 256
  want = args.shard_slots
  # every rank must end up with the SAME slot count (the id stream's range is world * slots): agree on the smallest that allocates
  table, failures = None, []
  for slots in [want] + [int(want * f) for f in (0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.25)]:
    ok = 1
    try:
      table = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=slots, max_capacity=slots,
                              device=str(dev), dim=dim, evict_strategy=de.HkvEvictStrategy.LRU, name="bench_m1s_%d" % rank)
    except Exception as e:   # noqa: BLE001
      failures.append({"slots": slots, "error": str(e)[:160]})
      ok, table = 0, None
    if world > 1:
      t_ok = torch.tensor([ok], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
      dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
      ok = int(t_ok.item())
    if ok:
      break
    if table is not None:
      del table
      table = None
      torch.cuda.empty_cache()
  assert table is not None, failures
  n_total = slots * world
  gen = torch.Generator(device=dev).manual_seed(SEED + 17 * rank)
  chunk = 4_000_000 * min(world, 8)     # ranks per chunk: ~4 M keys of them are this rank's
  vcap = min(4_400_000, max(1024, int(n_total / world * 1.1) + 1024))
  vals_fill = (torch.randn((vcap, dim), generator=gen, device=dev) * 0.01).to(dtype)
  t0 = time.perf_counter()
  # coldest ranks first (an LRU table: the order of the bulk load is the order of the scores), this rank's keys only
  for lo in range(((n_total - 1) // chunk) * chunk + 1, 0, -chunk):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(n_total, lo + chunk - 1) + 1, dtype=torch.int64, device=dev))
    if world > 1:
      k = k[((k & 0x7FFFFFFF) % world) == rank]   # default_partition_fn (PY/dynamic_embedding_variable.py:165-197)
    for a in range(0, k.numel(), vcap):
      kk = k[a:a + vcap]
      table._table.upsert(kk, vals_fill[:kk.numel()], unique_keys=True)
  resident = int(table.size().item())
  t_fill = time.perf_counter() - t0
  del vals_fill
  capacity = table._table.capacity()
  new_ratio = args.new_key_ratio if args.new_key_ratio is not None else 0.0
  idf = IdFactory(torch, dev, B, n_total, new_ratio, n_total + 1 + rank * (1 << 40), SEED + 1000 * rank + 7)
  values = (torch.randn((B, dim), generator=gen, device=dev) * 0.01).to(dtype)
  verified = {}
  # one rank: the route driver with device copies ("local"); TFRA_BENCH_FORCE_A2A=1 under torch.distributed.run --nproc-per-node 1:
  # the REAL transport with a one-rank RCCL communicator pair (grouped ncclSend / ncclRecv to itself: what the collectives' launches cost)
  forced = world == 1 and os.environ.get("TFRA_BENCH_FORCE_A2A") == "1" and dist is not None and dist.is_initialized()
  transport = ("rccl" if dist.get_backend() == "nccl" else "staged") if forced else ("local" if world == 1 else "auto")
  m = routed_assign_measure(args, torch, dist, de, dev, world, rank, table, idf, values, transport, K, W, verified, "routed_step")
  table._table.check_errors()
  census = {k: int(v) for k, v in table._table.slot_census().items()}
  size_end = int(table.size().item())
  verified.update({"check_errors_clean": True, "size_le_capacity": size_end <= capacity, "no_locked_slot": census["locked"] == 0})
  bad = [k for k, v in verified.items() if v is False]
  if world > 1:
    allbad = [None] * world
    dist.all_gather_object(allbad, bad)
    bad = sorted(set(sum(allbad, [])))
  assert not bad, "bench verification failed: %s (route: %s)" % (bad, m["stats"])
  uniq = idf.keys(4)
  U = int(np.mean([torch.unique(uniq[i]).numel() for i in range(4)]))
  med, secs, host_s = m["med"], m["secs"], m["host_s"]
  ms = med / K * 1e3
  value = world * B * K / med
  prof = profile_summary()
  step_bytes = B * (8 + 2 * Rb) + U * (16 + 2 * Rb)                              # per GPU, the metric's definition
  owner_bytes = m["served_ids"] * (8 + 2 * Rb) + m["served_distinct"] * (16 + 2 * Rb)   # what the owner's launch moves for the ids it serves
  res = {
      "metric": "embedding lookup+insert pairs/s (dim=64 fp32, %d-GPU hash-sharded bounded table, %d slots per GPU, %d %% never-seen ids per batch: "
                "lookup + insert_or_assign with score-based eviction, ids / rows / values routed)" % (world, capacity, round(100 * new_ratio)),
      "value": value, "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
      "driver": "routed_overlapped_step",
      "config": {
          "workload": "BASELINE metric (dim 64 fp32, Zipf-1.2) on configs[3]'s sharding: %d shard(s) x %d slots (bounded Hkv LRU tables at capacity, "
                      "%d keys resident per GPU; owner = default_partition_fn), per-GPU batch=%d from the GLOBAL Zipf-1.2 over the %d ranks; step = "
                      "lookup(B) + insert_or_assign(B, last occurrence wins): distinct ids / rows / values routed (%s), the owner runs "
                      "tfra_table_step_overlap on the ids it serves" % (world, capacity, resident, B, n_total,
                                                                        "RCCL alltoall over xGMI" if world > 1 else "one rank through the route driver: device copies"),
          "slots": capacity, "requested_slots": want, "alloc_failures": failures, "resident_after_prefill": resident,
          "global_batch": B * world, "keys_per_gpu": resident, "new_key_ratio": new_ratio, "unique_keys_per_batch": U,
          "unique_ratio": round(U / B, 4), "prefill_s": round(t_fill, 1), "steps_per_host_call": 1,
          "parallelism": ("key-hash sharded x%d, %s" % (world, "RCCL alltoall over xGMI (the driver's own communicator pair)" if m["rccl_ranks"] else
                                                        "alltoall STAGED THROUGH THE HOST (gloo: a functional run, ranks sharing a GPU — not a measurement)")) if world > 1 else
                         ("single GPU through the route driver, transport %s (one-rank communicators)" % transport if forced else "single GPU through the route driver (no transport)"),
          "route": "assign_route", "rccl_ranks_seen": m["rccl_ranks"], "batches_fed_ahead": m["ahead"],
          "host_enqueue_ms_per_step": round(1e3 * host_s / K, 4),
          "served_ids_per_step": m["served_ids"], "served_distinct_ids_per_step": m["served_distinct"],
          "route_stats": m["stats"], "verified": verified,
          "launches_per_step": "critical path (caller's stream): gather(values) + alltoall + owner step launch + alltoall + gather(rows) = 3 kernels "
                               "+ 2 collectives; ahead, on the driver's own stream: 2 route-plan launches + 2 small collectives + 1 copy per batch",
          "timing": {"value": timing_note(secs, K)},
      },
      "roofline": {
          "bound": "hbm", "kernel": "step_k at the owner (tfra_table_step_overlap on the ids this rank serves: lookup + write-back in ONE launch)",
          "achieved": owner_bytes / m["owner_launch_us"] / 1e3 if m["owner_launch_us"] else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": owner_bytes / m["owner_launch_us"] / 1e3 / HBM_PEAK_GBS if m["owner_launch_us"] else None,
          "traffic": traffic_of(prof, "m1s", "step_k") if world == 1 and not forced else None,
          "traffic_source": (prof or {}).get("_source") if (world == 1 and not forced and traffic_of(prof, "m1s", "step_k") is not None) else None,
          "algorithmic_bytes_per_launch": owner_bytes, "avg_launch_us": m["owner_launch_us"], "launches_timed": m["launches_timed"],
          "step_frac": step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "step_algorithmic_bytes": step_bytes,
          "step_bytes_definition": "per GPU: B*(8+2*Rb) for the lookup + U*(16+2*Rb) for the write-back of the batch's U distinct keys (SURVEY §8d)",
          "timing": "HIP events the owner's step driver records around its launch (they include one dispatch), 24 launches on fresh batches",
      },
  }
  del table
  import gc
  gc.collect()
  torch.cuda.empty_cache()
  return res



# ------------------------------------------------------------------ m1g: the metric's step on a GROWING table (TFRA's default creator)
def run_growing_assign(args, torch, de, dev):
  """lookup(B) + insert_or_assign(B) of Zipf-1.2 batches on a CuckooHashTable (no eviction, no max_capacity: what `CuckooHashTableCreator`,
  TFRA's default, instantiates) with `--keys` (10^8) resident keys, dim 64 fp32: round 6 — the overlapped step takes such tables too."""
  B, K, W = args.batch, args.steps, args.warmup
  dim, dtype, Rb = 64, torch.float32, 256
  n_keys = args.keys
  new_ratio = args.new_key_ratio if args.new_key_ratio is not None else 0.0
  table = de.CuckooHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), device=str(dev), dim=dim, name="bench_m1g", init_size=int(n_keys * 1.4))
  gen = torch.Generator(device=dev).manual_seed(SEED + 3)
  chunk = 4_000_000
  vals_fill = torch.randn((chunk, dim), generator=gen, device=dev) * 0.01
  for lo in range(1, n_keys + 1, chunk):
    k = keys_of_ranks_torch(torch, torch.arange(lo, min(n_keys, lo + chunk - 1) + 1, dtype=torch.int64, device=dev))
    table._table.upsert(k, vals_fill[:k.numel()], unique_keys=True)
  del vals_fill
  idf = IdFactory(torch, dev, B, n_keys, new_ratio, n_keys + 1, SEED + 11)
  values = torch.randn((B, dim), generator=gen, device=dev) * 0.01
  nsteps = W + WINDOWS * K
  ids = idf.keys(nsteps + 3)
  U = int(np.mean([torch.unique(ids[i]).numel() for i in range(4)]))
  outs = [torch.empty((B, dim), dtype=dtype, device=dev) for _ in range(2)]
  ovl = de.OverlapAssignStep(table)
  ovl.prime(ids[0])
  for i in range(W):
    ovl.step(values, ids[i + 1], ids[i + 2])
  runs = [ovl.make_run([ids[W + c]], [values], [outs[c & 1]], ids_after=ids[W + c + 1], values_before=values if (W + c) else None, ids_after2=ids[W + c + 2])
          for c in range(WINDOWS * K)]
  secs, med, host_s = timed_windows(torch, None, 1, dev, K, lambda i: runs[i - W](), first=W)
  ovl.flush()
  st = ovl.stats()
  last = ids[nsteps - 1]
  got, ex = table.lookup(last, return_exists=True)
  verified = {"growing_table_last_batch": bool(ex.all()) and bool(torch.equal(got, last_occurrence_rows(torch, last, values))),
              "growing_table_every_timed_step_overlapped": st["sequential"] <= W, "check_errors_clean": True}
  table._table.check_errors()
  bad = [k for k, v in verified.items() if v is False]
  assert not bad, "bench verification failed: %s (%s)" % (bad, st)
  ms = med / K * 1e3
  step_bytes = B * (8 + 2 * Rb) + U * (16 + 2 * Rb)
  res = {"metric": "embedding lookup+insert pairs/s (dim=64 fp32, GROWING table, %d resident keys, %d %% never-seen ids per batch)" % (n_keys, round(100 * new_ratio)),
         "value": B * K / med, "unit": "pairs/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": ms, "driver": "overlapped_step",
         "config": {"workload": "the metric's step on a growing table (CuckooHashTable: no eviction strategy, no max_capacity — TFRA's default creator), %d resident "
                                "keys, batch=%d Zipf-1.2, %d %% never-seen ids" % (n_keys, B, round(100 * new_ratio)),
                    "host_enqueue_ms_per_step": round(1e3 * host_s / K, 4), "unique_keys_per_batch": U, "verified": verified,
                    "overlapped_step_stats": st, "timing": {"value": timing_note(secs, K)}},
         "roofline": {"bound": "hbm", "kernel": "step_k_gen (the step's ONE launch, strategy read at run time)", "step_frac": step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "frac": step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "step_algorithmic_bytes": step_bytes}}
  del table, ovl, runs
  import gc
  gc.collect()
  torch.cuda.empty_cache()
  return res



# ------------------------------------------------------------------ c2 / c4: growing table behind de.Variable, fused optimizer
def run_sharded(args, torch, dist, de, dev, world, rank, cfg):
  """cfg 'c2' (configs[1], one GPU): 100 M keys, rows [p|m|v], 10 % never-seen ids, lookup + fused sparse Adam; look-ahead
  driver and plain calls.
  cfg 'c4' (configs[3], ANY number of GPUs, also 1): `--c4-keys` (5*10^8) keys PER GPU, hash-sharded by the reference's
  default partitioner over `world` shards, rows without optimizer slots (fused SGD: 5*10^8 rows [p|m|v] would not fit
  one GPU), per-GPU batch from the GLOBAL Zipf-1.2 over world*5*10^8 ranks, the whole step driven by tfra_route_* — at
  world 1 the same driver without a transport (device copies instead of alltoalls)."""
  from tfra_amd.dynamic_embedding.distributed import AllToAllEmbedding
  DIM = 64
  B, K, W = args.batch, args.steps, args.warmup
  c4 = cfg == "c4"
  n_local = args.c4_keys if c4 else args.keys
  n_total = n_local * world
  new_ratio = 0.0 if c4 else (args.new_key_ratio if args.new_key_ratio is not None else 0.1)
  nsteps = W + WINDOWS * K

  opt = de.optimizers.SGD(0.05) if c4 else de.optimizers.Adam(1e-3, 0.9, 0.999, 1e-8)
  deo = de.DynamicEmbeddingOptimizer(opt)
  # init_size: the resident keys plus room for the never-seen ids of every batch of the run (warm-up and both drivers), so
  # that the table does not grow inside a timed region (it grows as soon as its true size passes max_load_factor)
  headroom = int(2.2 * (nsteps + 8) * B * max(new_ratio, 0.0)) + (1 << 20)
  var = de.Variable(dim=DIM, devices=[str(dev)], name="bench_rank%d" % rank, initializer=0.0, init_size=int(n_local * 1.05) + headroom,
                    **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  force_a2a = os.environ.get("TFRA_BENCH_FORCE_A2A") == "1" and dist.is_initialized()
  table = var.tables[0]
  gen = torch.Generator(device=dev).manual_seed(SEED + rank)
  chunk = 4_000_000
  t_fill = time.perf_counter()
  fill_vals = torch.randn((chunk, DIM), generator=gen, device=dev) * 0.01
  for lo in range(1, n_total + 1, chunk):
    r = torch.arange(lo, min(n_total, lo + chunk - 1) + 1, dtype=torch.int64, device=dev)
    k = keys_of_ranks_torch(torch, r)
    if world > 1:
      k = k[((k & 0x7FFFFFFF) % world) == rank]   # default_partition_fn (PY/dynamic_embedding_variable.py:165-197)
    table._table.upsert(k, fill_vals[:k.numel()], unique_keys=True)
  del fill_vals
  resident = int(table.size().item())
  t_fill = time.perf_counter() - t_fill

  idf = IdFactory(torch, dev, B, n_total, new_ratio, n_total + 1 + rank * (1 << 40), SEED + 1000 * rank + 1)
  grads = torch.randn((B, DIM), generator=gen, device=dev) * 0.01
  tm = Timer(torch)
  single = not c4 and world == 1
  secs_plain = med_plain = None
  route, route_note, rs, rccl_ranks = None, None, None, None
  if single:
    ids = idf.keys(nsteps + 1)
    uniq_ratio = float(np.mean([torch.unique(ids[i]).numel() / B for i in range(4)]))
    prefetch = de.PrefetchStep(var, deo).prime(ids[0])
    for i in range(W):
      prefetch.step(grads, ids[i + 1])
    secs, med, host_s = timed_windows(torch, dist, world, dev, K, lambda i: prefetch.step(grads, ids[i + 1]), first=W)
    del prefetch
    ids = idf.keys(nsteps)

    def plain(i):
      var.lookup(ids[i])
      deo.apply_sparse(var, ids[i], grads)

    for i in range(W):
      plain(i)
    secs_plain, med_plain, _ = timed_windows(torch, dist, world, dev, K, plain, first=W)
  else:
    # N >= 1 through the route: the id-only half — distinct ids, owner-major order, count exchange, id alltoall, both
    # de-duplication plans — runs up to three batches ahead on the driver's own streams (tfra_route_*: three C calls per step,
    # grouped ncclSend/ncclRecv on its own RCCL communicators); per step: find -> alltoall(rows) -> gather, gradient sums ->
    # alltoall(grads) -> fused update at the owner.  Should the C driver not come up on some rank, every rank falls back to the
    # same sequence driven from Python (RoutedPrefetchStep); recorded in config.route.
    from tfra_amd.dynamic_embedding.distributed import NativeRoutedStep, RoutedPrefetchStep
    route = os.environ.get("TFRA_BENCH_ROUTE", "native")
    ahead = int(os.environ.get("TFRA_ROUTE_AHEAD", "3"))   # batches whose ids are known before their step (input pipeline)
    Wr = max(W, 8)   # the routed pipeline runs three batches ahead on its own streams: the first steps fill it (round 3: 5 warm-up steps, first window 136 us against 104)
    ids = idf.keys(Wr + WINDOWS * K + ahead + 1)
    uniq_ratio = float(np.mean([torch.unique(ids[i]).numel() / B for i in range(4)]))
    if route == "native":
      err = None
      try:
        rs = NativeRoutedStep(var, deo, partition_mode=0, force_collectives=force_a2a, max_batch=B,
                              threaded=os.environ.get("TFRA_ROUTE_THREAD", "1") != "0")
      except Exception as e:   # noqa: BLE001 — agreed on below
        err = e
      if dist.is_initialized():
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        agreed = int(ok.item()) == 1
      else:
        agreed = err is None
      if not agreed:
        if rs is not None:
          rs.close()
        rs, route = None, "prefetch"
        route_note = "native route driver unavailable (%s): fell back to the Python-driven route" % (str(err)[:200] if err else "on another rank")
      if rank == 0:
        print("[bench] route: %s, world %d%s" % (route, world, "" if route_note is None else " — " + route_note), file=sys.stderr, flush=True)
    if route != "native":
      if not dist.is_initialized():
        raise RuntimeError("the Python-driven route needs torch.distributed (run under torch.distributed.run)")
      rs = RoutedPrefetchStep(var, deo, partition_mode=0, force_collectives=force_a2a)
      ahead = 2
    for j in range(ahead):
      rs.feed(ids[j])

    def routed(i):
      out = rs.lookup()
      rs.apply(grads)
      rs.feed(ids[i + ahead])
      return out

    rccl_ranks = getattr(rs, "rccl_ranks", None)
    for i in range(Wr):
      routed(i)
    secs, med, host_s = timed_windows(torch, dist, world, dev, K, routed, first=Wr)
    for _ in range(ahead):   # drain the batches fed ahead
      rs.lookup(); rs.apply(grads)
    torch.cuda.synchronize()
    if hasattr(rs, "close"):
      rs.close()
  size_after = int(table.size().item())
  table._table.check_errors()

  # ---- per-kernel timings, each phase alone -----------------------------------------------------------------
  rc = raw_calls(torch, dev)
  h = table._table._h
  out_buf = torch.empty((B, DIM), dtype=torch.float32, device=dev)
  dflt_row = table._default_value
  kids = idf.keys(16)
  finds = [rc.find(h, kids[j], out_buf, dflt_row) for j in range(8)]
  find_us = tm.us(lambda i: finds[i % 8]())
  find_b2b_us = tm.us(lambda i: finds[0]())
  find_bytes = B * (8 + 2 * DIM * 4)
  p = opt.params(1)
  dflt = table._default_value.to(torch.float32)
  grad_half_us = plan_us = None
  U = int(torch.unique(kids[8]).numel())
  if de.DynamicEmbeddingOptimizer.can_plan(var, B):
    plans = [de.table_ops.SparsePlan(dev, DIM) for _ in range(6)]
    for j, pl in enumerate(plans):
      pl.build(kids[8 + j], sync=False)
    torch.cuda.synchronize()
    counts = plans[0].read()[0]
    U = counts["many"] + counts["few"]
    halves = [rc.apply_planned(h, p, plans[j], grads, dflt) for j in range(6)]
    grad_half_us = tm.us(lambda i: halves[i % 6](), reps=30)
    builds = [rc.plan_build(plans[j], kids[8 + j], DIM) for j in range(6)]
    plan_us = tm.us(lambda i: builds[i % 6](), reps=30)
  nslot = 0 if c4 else 2
  per_key = 8 + (2 + 2 * nslot + 1) * DIM * 4          # key + gradient + (1+S) rows read and written
  wb_bytes = B * (8 + DIM * 4) + U * per_key           # ids + gradient rows once, fused update on the unique keys
  uniq = torch.unique(kids[14])
  gsum = torch.randn((uniq.numel(), DIM), generator=gen, device=dev) * 0.01
  apply_one = rc.apply_optimizer(h, p, uniq, gsum, dflt)
  apply_us = tm.us(lambda i: apply_one(), reps=30)
  apply_bytes = int(uniq.numel()) * per_key
  prof = profile_summary()

  ms = med / K * 1e3
  value = world * B * K / med
  step_bytes = find_bytes + wb_bytes
  oname = "SGD" if c4 else "Adam"
  res = {
      "metric": "embedding lookup+insert pairs/s (dim=64 fp32, Zipf-1.2, lookup + fused sparse-%s write-back%s)"
                % (oname, ", %d-GPU hash-sharded table, ids/rows/grads routed" % world if c4 else ""),
      "value": value, "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
      "config": {
          "workload": ("BASELINE configs[3]: %d shard(s) x %d resident keys (%d total, owner = default_partition_fn), dim=64 fp32 rows, "
                       "per-GPU batch=%d from the GLOBAL Zipf-1.2 over all keys, routed lookup + routed fused SGD write-back"
                       % (world, resident, n_total, B)) if c4 else
                      ("BASELINE configs[1]: %d resident keys, dim=64 fp32 rows [p|m|v], Zipf-1.2 batch=%d with %d %% never-seen keys per "
                       "batch, lookup + dedup + fused sparse Adam (insert/write-back)" % (resident, B, round(100 * new_ratio))),
          "global_batch": B * world, "keys_per_gpu": resident, "keys_per_gpu_after_timed_steps": size_after,
          "new_key_ratio": new_ratio, "unique_ratio": round(uniq_ratio, 4), "unique_keys_per_batch": U,
          "parallelism": ("key-hash sharded x%d, RCCL alltoall" % world) if world > 1 else ("single GPU through the route driver (no transport)" if c4 else "single GPU"),
          "multi_rank_rccl_note": None if not c4 else "no RCCL communicator with more than one rank has been formed on the builder's side (one GPU per box): "
                                                      "the N>1 numbers are the driver's to measure; nothing is projected here",
          "table_ops_per_s": world * (B + U) * K / med, "table_ops_per_s_counts": "B lookups + U fused row updates (the distinct keys of the batch) per GPU and step",
          "prefill_s": round(t_fill, 1),
          "host_enqueue_ms_per_step": round(1e3 * host_s / K, 4), "route": route, "route_note": route_note,
          "rccl_ranks_seen": rccl_ranks,   # ncclCommCount of the route driver's own communicators (None: no RCCL transport — one rank, or gloo-staged)
          "timing": {"value": timing_note(secs, K)},
          "drivers": {
              "value": ("tfra_table_step_prefetch: ONE C call per step = lookup + hot sums + fused Adam of batch i on the main "
                        "stream, CSR-by-key plan of batch i+1 on a second stream; one plan built per step inside the timed region")
              if single else {
                  "native": "tfra_route_* (C driver; grouped ncclSend/ncclRecv at world > 1, device copies at world 1): id-only half of "
                            "the alltoall route up to three batches ahead on its own streams; per step find -> alltoall(rows) -> gather, "
                            "gradient sums -> alltoall(grads) -> fused update at the owner",
                  "prefetch": "RoutedPrefetchStep (the same sequence driven from Python through torch.distributed)"}.get(route),
              "value_plain_call": "tfra_table_find then tfra_table_apply_sparse (plan built inside the call): the reference's op "
                                  "sequence lookup -> optimizer apply, no look-ahead" if single else None},
      },
      "roofline": {
          "bound": "hbm", "kernel": "find_kernel<16,4> (embedding lookup, default fill fused)",
          "achieved": find_bytes / find_us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": find_bytes / find_us / 1e3 / HBM_PEAK_GBS, "traffic": traffic_of(prof, cfg, "find_kernel"),
          "traffic_source": (prof or {}).get("_source") if traffic_of(prof, cfg, "find_kernel") is not None else None,
          "algorithmic_bytes_per_launch": find_bytes, "avg_launch_us": find_us,
          "avg_launch_us_same_batch_back_to_back": find_b2b_us,
          "step_frac": step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if world == 1 else None,
          "step_algorithmic_bytes": step_bytes,
          "step_bytes_definition": "B*(8+2*Rb) lookup + B*(8+Rb) ids and gradient rows + U*(8+(3+2S)*Rb) fused update on the U unique keys (S optimizer slots)",
          "by_survey_pair_count": B * 1048 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if world == 1 else None,
          "timing": "HIP events on the launching stream around 30-50 launches, a different batch each",
          "kernels": {
              "hot_sums_kernel + apply_csr_kernel<%s> (gradient half: duplicate sums + fused update)" % oname.upper(): {
                  "avg_launch_us": grad_half_us, "algorithmic_bytes_per_launch": wb_bytes, "unique_keys": U,
                  "achieved_GBps": wb_bytes / grad_half_us / 1e3 if grad_half_us else None,
                  "frac": wb_bytes / grad_half_us / 1e3 / HBM_PEAK_GBS if grad_half_us else None,
                  "traffic": (traffic_of(prof, cfg, "hot_sums_kernel") or 0) + (traffic_of(prof, cfg, "apply_csr_kernel") or 0) or None},
              "csr_tile_kernel + csr_bucket_kernel + csr_scatter_kernel (id-only plan, second stream)": {"avg_launch_us": plan_us},
              "apply_kernel<%s> alone on pre-summed unique keys" % oname.upper(): {
                  "avg_launch_us": apply_us, "algorithmic_bytes_per_launch": apply_bytes,
                  "achieved_GBps": apply_bytes / apply_us / 1e3, "frac": apply_bytes / apply_us / 1e3 / HBM_PEAK_GBS},
          },
      },
  }
  if med_plain is not None:
    res["value_plain_call"] = B * K / med_plain
    res["ms_per_step_plain_call"] = med_plain / K * 1e3
    res["roofline"]["step_frac_plain_call"] = step_bytes / (med_plain / K) / 1e9 / HBM_PEAK_GBS
    res["config"]["timing"]["value_plain_call"] = timing_note(secs_plain, K)
  del var, table, deo
  import gc
  gc.collect()
  torch.cuda.empty_cache()
  return res


# ------------------------------------------------------------------ c5: 26 tables on one GPU, FTRL
def run_c5(args, torch, de, dev):
  """BASELINE configs[4] on ONE GPU: 26 tables, key counts log-spaced and scaled so that the rows [p|accum|linear] fill
  most of the HBM, dims cycling {16, 32, 64, 128}, one id per table per sample (batch B per table), combined lookup +
  FTRL write-back.  One C call per step for all tables (tfra_multi_step_prefetch) vs one PrefetchStep call per table vs
  plain per-table calls."""
  NT, B, K, W = 26, args.batch, max(10, args.steps // 4), max(6, args.warmup // 4)   # (>= 6 warm-up steps: the per-table drivers rotate four plan objects, each allocates at its first use)
  dims = [16, 32, 64, 128]
  free, _ = torch.cuda.mem_get_info()
  budget = 0.70 * free   # bytes for the rows of all tables at load factor <= 0.75
  shape = np.logspace(7.0, 9.0, NT)          # 10 M ... 1 B before scaling
  row_bytes = np.array([3 * dims[i % 4] * 4 + 8.6 for i in range(NT)])
  scale = budget * 0.70 / float(np.sum(shape * row_bytes))
  sizes = np.maximum((shape * scale).astype(np.int64), 100_000)
  rng = np.random.default_rng(SEED + 5)
  opt = de.optimizers.Ftrl(0.05, l1_regularization_strength=1e-3, l2_regularization_strength=1e-3)
  deo = de.DynamicEmbeddingOptimizer(opt)
  tabs, ids, grads = [], [], []
  gen = torch.Generator(device=dev).manual_seed(SEED)
  t0 = time.perf_counter()
  for i in range(NT):
    d, n = dims[i % 4], int(sizes[i])
    v = de.Variable(dim=d, name="c5_%d" % i, initializer=0.0, init_size=int(n * 1.05), devices=[str(dev)],
                    **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    zero = torch.zeros((4_000_000, d), device=dev)
    for lo in range(1, n + 1, 4_000_000):
      k = keys_of_ranks_torch(torch, torch.arange(lo, min(n, lo + 3_999_999) + 1, dtype=torch.int64, device=dev) + (i << 40))
      v.tables[0]._table.upsert(k, zero[:k.numel()], unique_keys=True)
    del zero
    tabs.append(v)
    ranks = zipf_bounded(rng, 4 * B, n).reshape(4, B)
    ids.append([keys_of_ranks_torch(torch, torch.from_numpy(ranks[j]).to(dev) + (i << 40)) for j in range(4)])
    grads.append(torch.randn((B, d), generator=gen, device=dev) * 0.01)
  torch.cuda.synchronize()
  t_fill = time.perf_counter() - t0
  total_keys = int(sum(int(v.size().item()) for v in tabs))

  def timed(step, sync):
    for s_ in range(W):
      step(s_)
    sync()
    t1 = time.perf_counter()
    for s_ in range(K):
      step(W + s_)
    sync()
    return (time.perf_counter() - t1) / K

  ms = de.MultiTablePrefetchStep(tabs, deo, streams=args.c5_streams, workers=args.c5_workers).prime([x[0] for x in ids])
  t_multi = timed(lambda s_: ms.step(grads, [x[(s_ + 1) & 3] for x in ids]), lambda: (ms.synchronize(), torch.cuda.synchronize()))
  pss = [de.PrefetchStep(v, deo).prime(ids[i][0]) for i, v in enumerate(tabs)]

  def per_table(s_):
    for i, ps in enumerate(pss):
      ps.step(grads[i], ids[i][(s_ + 1) & 3])
  t_pre = timed(per_table, torch.cuda.synchronize)

  def plain(s_):
    p = deo.begin_step()
    for i, v in enumerate(tabs):
      v.lookup(ids[i][s_ & 3])
      deo.apply_sparse(v, ids[i][s_ & 3], grads[i], p)
  t_plain = timed(plain, torch.cuda.synchronize)
  Rb = np.array([dims[i % 4] * 4 for i in range(NT)], dtype=np.float64)
  uniq = np.array([float(torch.unique(ids[i][0]).numel()) for i in range(NT)])
  step_bytes = float(np.sum(B * (8 + 2 * Rb) + B * (8 + Rb) + uniq * (8 + 7 * Rb)))
  res = {
      "metric": "embedding lookup+insert pairs/s (26 tables on one GPU, dims {16,32,64,128}, Zipf-1.2, lookup + fused FTRL write-back)",
      "value": NT * B / t_multi, "unit": "pairs/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": t_multi * 1e3,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
      "value_one_call_per_table": NT * B / t_pre, "ms_per_step_one_call_per_table": t_pre * 1e3,
      "value_plain_call": NT * B / t_plain, "ms_per_step_plain_call": t_plain * 1e3,
      "config": {
          "workload": "BASELINE configs[4] on one GPU: 26 tables, %d resident keys in total (%d ... %d per table, log-spaced, scaled to "
                      "%.0f GB of rows [p|accum|linear]), dims cycling {16,32,64,128} fp32, batch=%d ids per table and step (Zipf-1.2), "
                      "lookup + duplicate sums + fused FTRL" % (total_keys, int(sizes.min()), int(sizes.max()),
                                                                float(np.sum(sizes * row_bytes)) / 1e9, B),
          "tables": NT, "global_batch": NT * B, "prefill_s": round(t_fill, 1), "streams": args.c5_streams, "host_threads": args.c5_workers,
          "drivers": {"value": "tfra_multi_step_prefetch: ONE C call per step for all 26 tables (host-thread pool, stream pairs round-robin)",
                      "value_one_call_per_table": "tfra_table_step_prefetch per table from one Python thread",
                      "value_plain_call": "Find + apply_sparse per table (the reference's op sequence)"}},
      "roofline": {"bound": "hbm", "kernel": "whole step (26 tables x find + hot_sums + apply_csr<FTRL>)", "achieved": step_bytes / t_multi / 1e9,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_bytes / t_multi / 1e9 / HBM_PEAK_GBS, "traffic": None,
                   "step_algorithmic_bytes": step_bytes, "step_frac": step_bytes / t_multi / 1e9 / HBM_PEAK_GBS},
  }
  return res


def run_c5_routed(args, torch, dist, de, dev, world, rank, budget_frac=0.5):
  """BASELINE configs[4] through the route, ANY number of GPUs (also 1): 26 tables, each hash-sharded over the ranks by the
  reference's default partitioner; per GPU the key counts of run_c5 (log-spaced, scaled to `budget_frac` of the HBM), dims cycling
  {16, 32, 64, 128}, per-GPU batch B per table from the GLOBAL Zipf-1.2 over the table's world x keys, fused FTRL at the owner.
  One `tfra_route` per table on ONE shared transport (one pair of RCCL communicators), driven by MultiTableRoutedStep's three
  multi-table calls per step (feed / lookup / apply).  Weak scaling: per-GPU work is fixed as N grows."""
  from tfra_amd.dynamic_embedding.distributed import MultiTableRoutedStep
  NT, B, K, W = 26, args.batch, max(6, args.steps // 4), max(4, args.warmup // 2)
  dims = [16, 32, 64, 128]
  free, _ = torch.cuda.mem_get_info()
  shape = np.logspace(7.0, 9.0, NT)
  row_bytes = np.array([3 * dims[i % 4] * 4 + 8.6 for i in range(NT)])
  scale = budget_frac * free * 0.70 / float(np.sum(shape * row_bytes))
  sizes = np.maximum((shape * scale).astype(np.int64), 100_000)     # keys PER GPU and table
  opt = de.optimizers.Ftrl(0.05, l1_regularization_strength=1e-3, l2_regularization_strength=1e-3)
  deo = de.DynamicEmbeddingOptimizer(opt)
  gen = torch.Generator(device=dev).manual_seed(SEED + rank)
  rng = np.random.default_rng(SEED + 5 + 1000 * rank)
  tabs, ids, grads = [], [], []
  t0 = time.perf_counter()
  NB = 8
  for i in range(NT):
    d, n_local = dims[i % 4], int(sizes[i])
    n_total = n_local * world
    v = de.Variable(dim=d, name="c5r_%d_r%d" % (i, rank), initializer=0.0, init_size=int(n_local * 1.1) + (1 << 16), devices=[str(dev)],
                    **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
    zero = torch.zeros((4_000_000, d), device=dev)
    for lo in range(1, n_total + 1, 4_000_000):
      k = keys_of_ranks_torch(torch, torch.arange(lo, min(n_total, lo + 3_999_999) + 1, dtype=torch.int64, device=dev) + (i << 40))
      if world > 1:
        k = k[((k & 0x7FFFFFFF) % world) == rank]   # default_partition_fn (PY/dynamic_embedding_variable.py:165-197)
      v.tables[0]._table.upsert(k, zero[:k.numel()], unique_keys=True)
    del zero
    tabs.append(v)
    ranks = zipf_bounded(rng, NB * B, n_total).reshape(NB, B)
    ids.append([keys_of_ranks_torch(torch, torch.from_numpy(ranks[j]).to(dev) + (i << 40)) for j in range(NB)])
    grads.append(torch.randn((B, d), generator=gen, device=dev) * 0.01)
  torch.cuda.synchronize()
  t_fill = time.perf_counter() - t0
  keys_per_gpu = int(sum(int(v.size().item()) for v in tabs))
  force_a2a = os.environ.get("TFRA_BENCH_FORCE_A2A") == "1" and dist.is_initialized()
  ms = MultiTableRoutedStep(tabs, deo, partition_mode=0, force_collectives=force_a2a, max_batch=B)
  ahead = 3
  for j in range(ahead):
    ms.feed([x[j % NB] for x in ids])

  def step(i):
    out = ms.lookup()
    ms.apply(grads)
    ms.feed([x[(i + ahead) % NB] for x in ids])
    return out

  for i in range(W):
    step(i)
  secs, med, host_s = timed_windows(torch, dist, world, dev, K, step, first=W, windows=3)
  for _ in range(ahead):
    ms.lookup(); ms.apply(grads)
  torch.cuda.synchronize()
  rccl_ranks = ms.rccl_ranks
  ms.close()
  for v in tabs:
    v.tables[0]._table.check_errors()
  Rb = np.array([dims[i % 4] * 4 for i in range(NT)], dtype=np.float64)
  uniq = np.array([float(torch.unique(ids[i][0]).numel()) for i in range(NT)])
  step_bytes = float(np.sum(B * (8 + 2 * Rb) + B * (8 + Rb) + uniq * (8 + 7 * Rb)))
  t_step = med / K
  res = {
      "metric": "embedding lookup+insert pairs/s (26 hash-sharded tables per GPU, dims {16,32,64,128}, Zipf-1.2, routed lookup + routed fused FTRL)",
      "value": world * NT * B / t_step, "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": t_step * 1e3,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
      "config": {
          "workload": "BASELINE configs[4]: 26 tables x %d shard(s), %d resident keys per GPU in total (%d ... %d per table and GPU, log-spaced), "
                      "dims cycling {16,32,64,128} fp32 rows [p|accum|linear], per-GPU batch=%d ids per table and step from the GLOBAL Zipf-1.2, "
                      "routed lookup + routed fused FTRL (one tfra_route per table, one shared transport)"
                      % (world, keys_per_gpu, int(sizes.min()), int(sizes.max()), B),
          "tables": NT, "global_batch": NT * B * world, "keys_per_gpu": keys_per_gpu, "prefill_s": round(t_fill, 1),
          "parallelism": ("26 tables, each key-hash sharded x%d, RCCL alltoall (shared communicators)" % world) if world > 1
                         else "single GPU through the route driver (no transport)",
          "route": "native", "rccl_ranks_seen": rccl_ranks, "host_enqueue_ms_per_step": round(1e3 * host_s / K, 4),
          "timing": {"value": timing_note(secs, K)},
          "driver": "MultiTableRoutedStep: tfra_route_feed / _lookup / _apply per table from three multi-table calls per step"},
      "roofline": {"bound": "hbm", "kernel": "whole step (26 tables x routed find + gradient sums + apply_csr<FTRL>)", "achieved": step_bytes / t_step / 1e9,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_bytes / t_step / 1e9 / HBM_PEAK_GBS, "traffic": None,
                   "algorithmic_bytes_per_launch": step_bytes, "avg_launch_us": t_step * 1e6,
                   "step_algorithmic_bytes": step_bytes, "step_frac": step_bytes / t_step / 1e9 / HBM_PEAK_GBS},
  }
  del ms, tabs
  import gc
  gc.collect()
  torch.cuda.empty_cache()
  return res


# ------------------------------------------------------------------ the ONE line the driver parses: < 6 KB
LINE_LIMIT = 6000


def _num(x, sig=5):
  """numbers short enough for a compact line: ints stay, floats keep `sig` significant digits"""
  if isinstance(x, bool) or x is None or isinstance(x, int):
    return x
  if isinstance(x, float):
    if x != x or x in (float("inf"), float("-inf")):
      return None
    return float("%.*g" % (sig, x))
  return x


def _pick(d, keys):
  return {k: _num(d[k]) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def _short(sv, n):
  sv = "" if sv is None else str(sv)
  return sv if len(sv) <= n else sv[: n - 1] + "…"


def compact_line(res, detail_path=None):
  """The bench line the driver keeps (its stdout tail is ~8 KB): value, roofline, cpu_baseline, scaling_point and ONE row per
  secondary workload / driver variant; everything else — per-driver prose, timing windows, verification flags, per-op CPU tables,
  per-kernel tables — goes to bench_detail.json (`detail`) and stderr."""
  cfg, rf, cb = res.get("config", {}), res.get("roofline", {}), res.get("cpu_baseline")
  line = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                     "dtype", "data"))
  line["metric"] = _short(res.get("metric"), 200)
  c = _pick(cfg, ("slots", "global_batch", "unique_ratio", "unique_keys_per_batch", "new_key_ratio", "steps_per_host_call",
                  "host_enqueue_ms_per_step", "resident_after_prefill", "keys_per_gpu", "parallelism", "route", "tables", "rccl_ranks_seen",
                  "served_ids_per_step", "served_distinct_ids_per_step", "batches_fed_ahead"))
  c["workload"] = _short(cfg.get("workload"), 360)
  c["driver"] = res.get("driver") or cfg.get("driver")
  if res.get("driver_rule"):
    c["driver_rule"] = _short(res["driver_rule"], 170)
  line["config"] = c
  # the run's own untimed verification (config.verified: every flag in bench_detail.json): ONE bit here — the AND of all of them over the
  # top-level workload and every secondary one that ran
  flags = dict(cfg.get("verified") or {})
  for name, rr in (res.get("secondary") or {}).items():
    for k, v in ((rr.get("config") or {}).get("verified") or {}).items():
      flags["%s.%s" % (name, k)] = v
  bools = [v for v in flags.values() if isinstance(v, bool)]
  line["verified"] = bool(bools) and all(bools)
  line["verified_flags"] = len(bools)
  r = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_profile_summary", "algorithmic_bytes_per_launch",
                 "avg_launch_us", "step_frac", "step_algorithmic_bytes"))
  if r.get("traffic_profile_summary") is None:
    r.pop("traffic_profile_summary", None)
  if "avg_launch_us" in r:
    r["avg_launch_us_is"] = "HIP events around the launch (incl. one dispatch)"
  r["kernel"] = _short(rf.get("kernel"), 150)
  if r.get("traffic") is not None and r.get("algorithmic_bytes_per_launch"):
    r["traffic_over_algorithmic"] = _num(r["traffic"] / r["algorithmic_bytes_per_launch"], 3)
  line["roofline"] = r
  if cb is not None:
    b = _pick(cb, ("value", "unit", "cores", "kind", "resident_keys", "table_ops_only_pairs_per_s", "host_cores",
                   "prefill_keys_per_s_init_size_8192_growth_included", "small_table_legs_keys"))
    po = cb.get("per_op") or {}
    for k in ("find_unique_ids_ops_per_s", "insert_or_assign_ops_per_s", "find_with_repeats_ops_per_s", "prefill_keys_per_s_init_size_N"):
      if k in po:
        b[k] = _num(po[k])
    b["sample"] = _short(cb.get("sample"), 330)
    b.update({k: _num(v) if not isinstance(v, str) else v for k, v in cpu_baseline_1e9().items()})
    line["cpu_baseline"] = b
  sp = res.get("scaling_point")
  if sp:
    line["scaling_point"] = dict(_pick(sp, ("value", "ms_per_step", "n_gpus", "launches_per_step", "error")), workload=_short(sp.get("workload"), 200))
  variants = {}
  for k, v in res.items():   # one number per driver variant of the top-level workload
    if k.startswith("value_") and isinstance(v, (int, float)):
      variants[k[6:]] = _num(v, 4)
  if variants:
    line["variants_pairs_per_s"] = variants
  sec = {}
  for name, rr in (res.get("secondary") or {}).items():
    if "error" in rr:
      sec[name] = {"error": _short(rr["error"], 160)}
      continue
    row = _pick(rr, ("value", "ms_per_step", "driver"))
    rrf, rcf = rr.get("roofline", {}), rr.get("config", {})
    row.update({"step_frac": _num(rrf.get("step_frac"), 4), "kernel_frac": _num(rrf.get("frac"), 4),
                "kernel": _short(rrf.get("kernel"), 60), "host_enqueue_ms_per_step": _num(rcf.get("host_enqueue_ms_per_step")),
                "unique_keys_per_batch": rcf.get("unique_keys_per_batch")})
    for k in ("value_overlapped_step", "value_look_ahead_driver", "value_op_surface", "value_plain_call"):
      if k in rr:
        row[k[6:]] = _num(rr[k], 4)
    sec[name] = row
  if sec:
    line["secondary"] = sec
  if detail_path:
    line["detail"] = detail_path
  out = json.dumps(line, separators=(",", ":"), ensure_ascii=False)
  if len(out) > LINE_LIMIT:   # never lose the line: drop the optional blocks, largest first
    for k in ("variants_pairs_per_s", "secondary"):
      line.pop(k, None)
      out = json.dumps(line, separators=(",", ":"), ensure_ascii=False)
      if len(out) <= LINE_LIMIT:
        break
  return out


def emit(res):
  """bench_detail.json (everything) next to bench.py — and under gpurun_out/ when that exists, so that it comes back from a GPU
  box — and the compact line as the LAST line on stdout (nothing long goes to stderr either: a driver may keep one merged tail)."""
  detail = None
  dirs = [os.environ["TFRA_BENCH_DETAIL_DIR"]] if os.environ.get("TFRA_BENCH_DETAIL_DIR") else [ROOT, os.path.join(ROOT, "gpurun_out")]
  for d in dirs:
    try:
      if d == os.path.join(ROOT, "gpurun_out") and not os.path.isdir(d):
        continue
      with open(os.path.join(d, "bench_detail.json"), "w") as f:
        json.dump(res, f, indent=1)
      detail = detail or "bench_detail.json"
    except OSError:
      pass
  print("[bench] full result: %s" % (detail or "bench_detail.json could not be written"), file=sys.stderr, flush=True)
  import ctypes
  ctypes.CDLL(None).fflush(None)   # RCCL's version banner sits in the C stdio buffer: out before the JSON line, not after it
  print(compact_line(res, detail), flush=True)


# ------------------------------------------------------------------ main
def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--c5-streams", type=int, default=4)
  ap.add_argument("--c5-workers", type=int, default=1)
  ap.add_argument("--config", choices=["m1b", "m1s", "m1g", "c3", "c2", "c4", "c5"], default=None,
                  help="default: m1b on one GPU (the metric's own configuration; c3 / c2 / c4 / c5 as secondary), m1s per GPU for N>1 (the metric's "
                       "step on a hash-sharded table, ids / rows / values routed; --config m1s --gpus 1: the same through the route driver at one rank); "
                       "c4: configs[3] with the fused-SGD gradient route; c5 with --gpus N: 26 hash-sharded tables per GPU through the multi-table route")
  ap.add_argument("--slots", type=int, default=1_000_000_000, help="m1b / c3: slots of the bounded table")
  ap.add_argument("--shard-slots", type=int, default=500_000_000, help="m1s: slots of the bounded table PER GPU (configs[3]: 4 B keys over 8 GPUs)")
  ap.add_argument("--keys", type=int, default=100_000_000, help="c2: resident keys")
  ap.add_argument("--c4-keys", type=int, default=500_000_000, help="c4: resident keys PER GPU")
  ap.add_argument("--batch", type=int, default=131072, help="ids per GPU per step")
  ap.add_argument("--new-key-ratio", type=float, default=None, help="c2 / m1b: share of never-seen keys per batch")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-secondary", action="store_true", help="skip the secondary c3 / c2 / c4 measurements of the default invocation")
  args = ap.parse_args()

  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    # `python bench.py --gpus N` without a launcher: become one — re-run this command line under torch.distributed.run, one process per
    # GPU (rank 0 prints the line to the inherited stdout), and leave with its exit code
    import socket
    import subprocess
    with socket.socket() as sk:
      sk.bind(("127.0.0.1", 0))
      port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: %s" % (args.gpus, " ".join(cmd[1:8])), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(cmd, env=env))

  if args.config is None and args.gpus == 1 and "WORLD_SIZE" not in os.environ and os.environ.get("TFRA_BENCH_FORCE_A2A") != "1":
    args._live_traffic = live_traffic(args)   # (before this process allocates anything on the GPU)

  import torch
  import torch.distributed as dist
  import tfra_amd.dynamic_embedding as de

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if args.gpus > 1 or world > 1 or os.environ.get("TFRA_BENCH_FORCE_A2A") == "1":
    if world != args.gpus:
      raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` — it starts its own ranks — or "
                       "torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("TFRA_BENCH_BACKEND", "nccl")  # "gloo": smoke-test the N>1 path on ONE GPU
    if backend == "gloo":
      local_rank = 0
      dist.init_process_group("gloo")
    else:
      torch.cuda.set_device(local_rank)
      dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
  dev = torch.device("cuda", local_rank)
  torch.cuda.set_device(dev)
  cfg = args.config or ("m1b" if (world == 1 and not dist.is_initialized()) else "m1s")
  if cfg == "m1s":
    res = run_metric_sharded(args, torch, dist, de, dev, world, rank)
  elif cfg == "m1g":
    assert world == 1, "m1g is a single-GPU configuration"
    res = run_growing_assign(args, torch, de, dev)
    res.update({"higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"})
  elif cfg in ("c3", "m1b"):
    assert world == 1, "%s is a single-GPU configuration" % cfg
    if args.config is None and not args.no_secondary:
      args._growth = measure_growth(torch, de, dev, 128, torch.float16, args.slots // 4)   # reported under secondary.c3
    res = run_bounded(args, torch, de, dev, cfg)
    if not args.no_secondary and args.config is None:
      keep = ("metric", "value", "value_overlapped_step", "value_overlapped_step_4_steps_per_host_call", "value_look_ahead_driver", "driver", "driver_rule", "value_op_surface_fused_ops", "value_op_surface_host_read_first", "value_plain_call", "value_op_surface", "value_op_surface_find_first", "value_accum",
              "value_op_surface_table_ops_only", "ms_per_step",
              "ms_per_step_plain_call", "ms_per_step_op_surface", "config", "roofline")
      sec = {}
      note("m1b done; secondary workloads")
      for name, fn in (("m1g", lambda: run_growing_assign(args, torch, de, dev)),     # the metric's step on TFRA's DEFAULT table type (growing)
                       ("c3", lambda: run_bounded(args, torch, de, dev, "c3")),       # configs[2]
                       ("c2", lambda: run_sharded(args, torch, dist, de, dev, 1, 0, "c2")),   # configs[1]
                       ("c4", lambda: run_sharded(args, torch, dist, de, dev, 1, 0, "c4")),   # configs[3] at N=1: the first point of the N-GPU curve
                       ("c5", lambda: run_c5_routed(args, torch, dist, de, dev, 1, 0, budget_frac=0.2))):   # configs[4] at N=1 through the sharded multi-table route
        try:
          r = fn()
          sec[name] = {k: r[k] for k in keep if k in r}
          note("secondary %s done" % name)
        except Exception as e:   # noqa: BLE001 — a secondary measurement must not lose the line
          sec[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
      res["secondary"] = sec
      # what a `--gpus N` run measures PER GPU is configs[3] through the route driver — not the metric's configuration above: the
      # N = 1 point of that curve, at the top level so that a scaling record is built from like workloads
      rl = res.get("routed_local") or {}
      res["scaling_point"] = {
          "workload": "the metric's step on THIS table THROUGH the route driver at one rank (tfra_assign_route_*: what every `--gpus N` rank runs, "
                      "with device copies where the alltoalls would be): route plan ahead, per step gather + copy + owner step launch on the "
                      "distinct ids + copy + gather", "value": rl.get("value"), "ms_per_step": rl.get("ms_per_step"), "n_gpus": 1,
          "launches_per_step": 4, "error": rl.get("error"),
          "note": "`bench.py --gpus N` (N > 1) runs the metric's step per GPU on a hash-sharded table (run_metric_sharded: lookup(B) + "
                  "insert_or_assign(B), ids / rows / values routed, the owner running the overlapped step); at ONE rank the route is the identity "
                  "and that step IS the top-level `value`; this entry is the same one-rank step with the route's kernels forced on.  No RCCL "
                  "communicator with more than one rank has been formed on the builder's side (one GPU per box): nothing is projected."}
  elif cfg == "c5":
    if world == 1 and not dist.is_initialized():
      res = run_c5(args, torch, de, dev)     # the local drivers on one GPU ...
      import gc
      gc.collect()
      torch.cuda.empty_cache()
      try:                                     # ... and the N = 1 point of the routed, sharded form every `--gpus N --config c5` run measures
        rr = run_c5_routed(args, torch, dist, de, dev, 1, 0)
        res["value_routed"] = rr["value"]; res["ms_per_step_routed"] = rr["ms_per_step"]
        res["scaling_point"] = {"workload": rr["config"]["workload"], "value": rr["value"], "ms_per_step": rr["ms_per_step"], "n_gpus": 1}
      except Exception as e:   # noqa: BLE001
        res["scaling_point"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    else:
      res = run_c5_routed(args, torch, dist, de, dev, world, rank)
  elif cfg == "c2":
    assert world == 1, "c2 is a single-GPU configuration (N > 1 runs configs[3]: --config c4)"
    res = run_sharded(args, torch, dist, de, dev, 1, 0, "c2")
  else:
    res = run_sharded(args, torch, dist, de, dev, world, rank, "c4")
  if rank == 0:
    if not args.no_cpu_baseline and world == 1:   # (the CPU leg belongs to the one-GPU line: at N > 1 the other ranks would sit in the final barrier for its 45 s)
      res["cpu_baseline"] = cpu_baseline(args.batch)
    emit(res)
  if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
