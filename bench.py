"""bench.py — dynamic-embedding hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--keys N_KEYS] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic ids, configuration
BASELINE.json configs[1] ("1xMI355X: 100M keys, dim=64 fp32, Zipf-1.2 batch=131072,
lookup+insert+sparse-Adam"):

    forward : embedding lookup of the B Zipf ids  (find, default fill fused)         -> [B,64]
    backward: gradients [B,64] -> duplicate ids summed -> fused sparse Adam on the unique keys,
              which is also the write-back/insert of the batch's keys (rows are [p|m|v])

`value` = ids looked up AND written back per second (lookup+insert pairs/s), whole job.  Inputs
(id batches, gradients) are resident in HBM before the timed region.  N>1: one process per GPU,
tables sharded by key hash, ids/rows/grads routed with alltoall over RCCL (weak scaling: per-GPU
keys and batch fixed).  Prints ONE JSON line on rank 0.
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

DIM = 64
ZIPF_S = 1.2
SEED = 20250205
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


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


# ------------------------------------------------------------------ CPU baseline (reference engine)
def cpu_baseline(batch, budget_s=15.0):
  """The reference's CPU path timed on this box's host cores: cuckoohash_map.hh (oracle/_ref,
  compiled in place from the reference) driven through the reference's write-back sequence
  (PY/dynamic_embedding_optimizer.py:165-204): unique -> 3 finds -> dense Adam -> 3 upserts."""
  import oracle
  from oracle import optimizers as oopt
  kind = "reference" if oracle.available("reference") else "port"
  cores = os.cpu_count() or 1
  threads = cores if kind == "reference" else 1
  n_keys = 4_000_000
  rng = np.random.default_rng(SEED)
  tabs = [oracle.CpuTable(DIM, np.float32, kind=kind, init_size=n_keys, threads=threads) for _ in range(3)]
  chunk = 500_000
  for lo in range(1, n_keys + 1, chunk):
    r = np.arange(lo, min(n_keys, lo + chunk - 1) + 1, dtype=np.int64)
    k = keys_of_ranks(r)
    tabs[0].insert(k, (rng.standard_normal((k.size, DIM)) * 0.01).astype(np.float32))
    z = np.zeros((k.size, DIM), np.float32)
    tabs[1].insert(k, z); tabs[2].insert(k, z)
  grads = (rng.standard_normal((batch, DIM)) * 0.01).astype(np.float32)
  zero = np.zeros(DIM, np.float32)
  done, t_total, step = 0, 0.0, 0
  while t_total < budget_s and step < 200:
    ids = keys_of_ranks(zipf_bounded(rng, batch, n_keys))
    t0 = time.perf_counter()
    uniq, idx = np.unique(ids, return_inverse=True)
    g = np.zeros((uniq.size, DIM), np.float32)
    np.add.at(g, idx, grads)
    p = tabs[0].find(uniq, zero); m = tabs[1].find(uniq, zero); v = tabs[2].find(uniq, zero)
    step += 1
    p, m, v = oopt.adam(p, m, v, g, 1e-3, 0.9, 0.999, 1e-8, step)
    tabs[0].insert(uniq, p); tabs[1].insert(uniq, m); tabs[2].insert(uniq, v)
    t_total += time.perf_counter() - t0
    done += batch
  return {
      "value": done / t_total, "unit": "lookup+insert pairs/s", "cores": threads, "kind": kind,
      "sample": "%d steps of batch %d (Zipf-1.2 over %d resident keys, dim 64 fp32, unique->3 finds->numpy Adam->3 "
                "upserts, %.1f s of CPU work)" % (step, batch, n_keys, t_total),
  }


# ------------------------------------------------------------------ main
def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--keys", type=int, default=100_000_000, help="resident keys PER GPU")
  ap.add_argument("--batch", type=int, default=131072, help="ids per GPU per step")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--plan", choices=["off", "prefetch"], default="prefetch",
                  help="where the id-only half of the write-back (which ids repeat, summation order, unique keys) is built: "
                       "prefetch = on a second HIP stream for batch i+1 while step i runs (ids known one batch ahead, as an "
                       "input pipeline provides them; one C call per step; every step builds exactly one plan inside the "
                       "timed region); off = inside the write-back call (tfra_table_apply_sparse).  Single GPU only.")
  ap.add_argument("--graph", action="store_true", help="replay the step as one captured HIP graph (same kernels; "
                  "measured equal to eager launches: the step is bound by kernel boundaries, not by the host)")
  args = ap.parse_args()

  import torch
  import torch.distributed as dist
  import tfra_amd.dynamic_embedding as de
  from tfra_amd.dynamic_embedding.distributed import AllToAllEmbedding

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if args.gpus > 1 or world > 1 or os.environ.get("TFRA_BENCH_FORCE_A2A") == "1":
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
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
  B, K, W = args.batch, args.steps, args.warmup
  n_local = args.keys
  n_total = n_local * world

  # ---- table: rows [p|m|v] co-located, sized so that no rehash happens -------------------------
  opt = de.optimizers.Adam(1e-3, 0.9, 0.999, 1e-8)
  deo = de.DynamicEmbeddingOptimizer(opt)
  var = de.Variable(dim=DIM, devices=[str(dev)], name="bench_rank%d" % rank, initializer=0.0, init_size=int(n_local * 1.05),
                    **de.DynamicEmbeddingOptimizer.variable_kwargs(opt))
  # TFRA_BENCH_FORCE_A2A=1 (with WORLD_SIZE=1 under torch.distributed.run): keep the whole N>1 route,
  # collectives included, on one rank — measures the routing overhead a multi-GPU step adds
  force_a2a = os.environ.get("TFRA_BENCH_FORCE_A2A") == "1" and dist.is_initialized()
  emb = (AllToAllEmbedding(var, partition_mode=0, dedup=os.environ.get("TFRA_BENCH_DEDUP", "1") == "1",
                           force_collectives=force_a2a) if (world > 1 or force_a2a) else None)
  table = var.tables[0]

  # ---- pre-fill: every key this rank owns (ranks 1..n_total, owner = default_partition_fn) ------
  gen = torch.Generator(device=dev).manual_seed(SEED + rank)
  chunk = 4_000_000
  t_fill = time.perf_counter()
  for lo in range(1, n_total + 1, chunk):
    r = torch.arange(lo, min(n_total, lo + chunk - 1) + 1, dtype=torch.int64, device=dev)
    k = keys_of_ranks_torch(torch, r)
    if world > 1:
      k = k[((k & 0x7FFFFFFF) % world) == rank]
    v = torch.randn((k.numel(), DIM), generator=gen, device=dev) * 0.01
    table._table.upsert(k, v, unique_keys=True)
  resident = int(table.size().item())
  t_fill = time.perf_counter() - t_fill

  # ---- inputs resident in HBM: all id batches + one gradient buffer -----------------------------
  rng = np.random.default_rng(SEED + 1000 * rank)
  ids_np = keys_of_ranks(zipf_bounded(rng, (K + W) * B, n_total)).reshape(K + W, B)
  ids_all = torch.from_numpy(ids_np).to(dev)
  uniq_ratio = float(np.mean([np.unique(ids_np[i]).size / B for i in range(min(8, K + W))]))
  grads = torch.randn((B, DIM), generator=gen, device=dev) * 0.01

  ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
  ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
  ev_c = [torch.cuda.Event(enable_timing=True) for _ in range(K)]

  use_graph = world == 1 and args.graph
  captured = None
  if use_graph:
    captured = de.CapturedTrainStep(var, deo, B)
    captured.grads.copy_(grads)
    captured.capture(warmup_ids=ids_all[0])

  plan_mode = args.plan if (world == 1 and emb is None and not use_graph) else "off"
  prefetch = None
  if plan_mode == "prefetch":
    prefetch = de.PrefetchStep(var, deo).prime(ids_all[0])

  def step(i, timed_idx=None, fused_call=False):
    ids = ids_all[i]
    if prefetch is not None and not fused_call:
      # ONE C call: [main: lookup + run sums + fused Adam of batch i] + [second stream: plan of batch i+1]
      return prefetch.step(grads, ids_all[(i + 1) % (K + W)])
    if captured is not None:
      # one HIP-graph replay = lookup + tile-reduce + bucket-merge + fused Adam (+ the batch copy)
      return captured.step(ids)
    if timed_idx is not None:
      ev_a[timed_idx].record()
    if emb is None:
      out = var.lookup(ids)
    else:
      out = emb.lookup(ids)
    if timed_idx is not None:
      ev_b[timed_idx].record()
    if emb is None:
      deo.apply_sparse(var, ids, grads)
    else:
      emb.apply_gradients(deo, grads)
    if timed_idx is not None:
      ev_c[timed_idx].record()
    return out

  for i in range(W):
    step(i)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(K):
    step(W + i, i)
  host_enqueue_s = time.perf_counter() - t0   # host time to enqueue the K steps (== elapsed when host-bound)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  fused_call_ms = None
  if prefetch is not None:
    # secondary timed loop with the write-back as ONE fused call on one stream (--plan off): gives the
    # per-launch event timings of the roofline block and the unprefetched step time for comparison
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(K):
      step((W + i) % (K + W), i, fused_call=True)
    torch.cuda.synchronize()
    fused_call_ms = 1e3 * (time.perf_counter() - t1) / K
  if use_graph:
    fwd_ms = bwd_ms = None  # phases are inside one graph launch
  else:
    fwd_ms = float(np.mean([ev_a[i].elapsed_time(ev_b[i]) for i in range(K)]))
    bwd_ms = float(np.mean([ev_b[i].elapsed_time(ev_c[i]) for i in range(K)]))

  # ---- roofline inputs, measured live with HIP events on the stream the kernels run on ------------
  # (1) lookup kernel find_kernel<16,4>: algorithmic bytes per lookup = 8 (key) + Rb (row read) + Rb
  #     (out write) = 520 B at dim 64 fp32; one launch processes B ids.  Timed both
  #     inside the timed region (ev_a..ev_b brackets exactly that launch each step) and back-to-back.
  reps = 50
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  ids0 = ids_all[W]

  def timed(fn):
    for _ in range(5):
      fn()
    e0.record()
    for _ in range(reps):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

  find_b2b_us = timed(lambda: table.lookup(ids0))
  # the same launch over a DIFFERENT batch each time (as in a step: rows not warm from the previous launch);
  # 50 launches between two events => the events' own cost (~4 us around a single launch) is amortised.
  # This is the duration rocprofv3 reports for find_kernel inside the steps (profiles/).
  rot = [0]

  def find_rotating():
    rot[0] = (rot[0] + 1) % (K + W)
    table.lookup(ids_all[rot[0]])

  find_rot_us = timed(find_rotating)
  find_bytes = B * (8 + 2 * DIM * 4)
  find_evt_us = fwd_ms * 1e3 if (fwd_ms is not None and world == 1) else None
  find_us = find_rot_us
  # (2) write-back pipeline tile_reduce -> bucket_merge -> apply_kernel<INDIRECT> (one C-ABI call):
  #     algorithmic bytes = B*(8 + Rb) (ids + gradient rows read once) + U*(8 + 7*Rb) (fused Adam on
  #     the unique keys) — the dedup itself has no algorithmic traffic.
  uniq, idx, cnt = de.device_ops.unique(ids0)
  U = int(uniq.numel())
  p = opt.params(1)
  wb_us = timed(lambda: table._table.apply_sparse(p, ids0, grads, table._default_value))
  wb_bytes = B * (8 + DIM * 4) + U * (8 + 7 * DIM * 4)
  # (2b) the same write-back split as the default step runs it: gradient half (tile_sums -> bucket_sums ->
  #      apply_kernel<INDIRECT>, main stream) and id-only half (plan build, second stream), each alone
  grad_half_us = plan_us = None
  if world == 1 and de.DynamicEmbeddingOptimizer.can_plan(var, B):
    plan0 = deo.plan(var, ids0)
    torch.cuda.synchronize()
    dflt = table._default_value.to(torch.float32)
    grad_half_us = timed(lambda: table._table.apply_planned(p, plan0, grads, dflt, sync=False))
    plan_us = timed(lambda: plan0.build(ids0, sync=False))
  # (3) the fused optimizer kernel alone on pre-summed unique keys
  gsum = torch.randn((U, DIM), generator=gen, device=dev) * 0.01
  apply_us = timed(lambda: table._table.apply_optimizer(p, uniq, gsum, table._default_value))
  apply_bytes = U * (8 + 7 * DIM * 4)
  # measured HBM traffic of the same kernels (rocprofv3 PMC passes, profiles/rNN_summary.json)
  prof = None
  try:
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_summary.json"))
    if cands:
      prof = json.load(open(os.path.join(ROOT, "profiles", cands[-1])))
  except OSError:
    prof = None

  def traffic_of(kname):
    try:
      return prof["kernels"][kname]["hbm_bytes_per_launch_corrected"]
    except (TypeError, KeyError):
      return None

  if rank == 0:
    ms = elapsed / K * 1e3
    value = world * B * K / elapsed
    ach = find_bytes / (find_us * 1e-6) / 1e9
    res = {
        "metric": "embedding lookup+insert pairs/s (dim=64 fp32, Zipf-1.2, lookup + sparse-Adam write-back)",
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: %d resident keys/GPU (%d total), dim=64 fp32 rows [p|m|v], Zipf-1.2 "
                        "batch=%d/GPU, lookup + dedup + fused sparse Adam (insert/write-back)" % (resident, n_total, B),
            "global_batch": B * world, "keys_per_gpu": resident, "unique_ratio": round(uniq_ratio, 4),
            "parallelism": "key-hash sharded x%d, RCCL alltoall" % world if world > 1 else "single GPU",
            "table_ops_per_s": 2 * value, "prefill_s": round(t_fill, 1),
            "host_enqueue_ms_per_step": round(1e3 * host_enqueue_s / K, 4),
            "launch": "hipGraph replay" if use_graph else ("one C call per step, two streams" if prefetch is not None else "eager"),
            "write_back_plan": {"off": "built inside the write-back call (tfra_table_apply_sparse)",
                                "prefetch": "id-only half of batch i+1 built on a second HIP stream while step i runs "
                                            "(tfra_table_step_prefetch; one plan per step inside the timed region)"}[plan_mode],
            "ms_per_step_fused_call": fused_call_ms,
        },
        "roofline": {
            "bound": "hbm", "kernel": "find_kernel<16,4> (embedding lookup, default fill fused)",
            "achieved": find_bytes / (find_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": find_bytes / (find_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_of("find_kernel"),
            "algorithmic_bytes_per_launch": find_bytes, "avg_launch_us": find_us,
            "avg_launch_us_same_batch_back_to_back": find_b2b_us,
            "avg_launch_us_single_launch_between_events_in_step": find_evt_us,
            "timing": "HIP events around 50 launches, a different resident-table batch each (avg_launch_us); also one launch between two events inside the steps of the fused-call loop (includes ~4 us of event cost)",
        },
        "roofline_write_back": {
            "bound": "hbm", "kernel": "tile_reduce_kernel + bucket_merge_kernel + apply_kernel<INDIRECT> "
                                      "(duplicate-gradient reduction + fused sparse Adam as ONE C-ABI call, --plan off)",
            "split_for_the_default_step": {
                "gradient_half_us_alone": grad_half_us, "gradient_half_kernels": "tile_sums_kernel + bucket_sums_kernel + "
                "apply_kernel<INDIRECT> (main stream)", "plan_build_us_alone": plan_us,
                "plan_build_kernels": "tile_reduce_kernel<plan> + bucket_merge_kernel<plan> + plan_finish_kernel (second stream)",
                "achieved_GBps_gradient_half": (wb_bytes / (grad_half_us * 1e-6) / 1e9) if grad_half_us else None},
            "achieved": wb_bytes / (wb_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": wb_bytes / (wb_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "traffic": (sum(traffic_of(k) for k in ("tile_reduce_kernel", "bucket_merge_kernel", "apply_kernel"))
                        if all(traffic_of(k) for k in ("tile_reduce_kernel", "bucket_merge_kernel", "apply_kernel")) else None),
            "algorithmic_bytes_per_launch": wb_bytes, "avg_launch_us": wb_us, "unique_keys": U,
            "note": "latency/occupancy bound, not bandwidth bound: per-kernel split in profiles/",
        },
        "phases": {
            "forward_ms": fwd_ms, "backward_ms": bwd_ms,
            "apply_kernel_alone": {"avg_launch_us": apply_us, "algorithmic_bytes_per_launch": apply_bytes,
                                   "achieved_GBps": apply_bytes / (apply_us * 1e-6) / 1e9},
        },
    }
    if not args.no_cpu_baseline:
      res["cpu_baseline"] = cpu_baseline(B)
    print(json.dumps(res))
  if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
