"""GPU: the overlapped step (tfra_table_step_overlap, csrc/tfra_step_impl.h) — lookup of batch i+1 in the same launch as the
write-back of batch i — must return exactly what the ops return one after the other (the reference's order:
HkvHashTableOfTensorsGpu::Find / ::Insert, K/hkv_hashtable_op_gpu.cu.cc:182-290; Insert exclusive, Find shared).

The check that needs no model of the eviction order: when a step call has run, the write-back of the PREVIOUS batch is
complete and the write-back of this batch has not started, so the table is exactly in the state this batch's lookup must
reflect — a plain tfra_table_find of the same ids (read-only, no scores touched) must return the same rows and exists flags,
bit for bit.  On top of that: a dictionary oracle where nothing is evicted, forced conflicts (keys the write-back evicts
while the next lookup asks for them), sentinel keys, repeats, steps without look-ahead, many steps from one host call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
  import torch
  import tfra_amd.dynamic_embedding as de
  return torch, de


def make_dense_table(torch, de, cap, dim, fill_keys, name, dtype=None):
  """bounded LRU table at max_capacity, pre-filled, and known to the host as dense (> 60 % of the slots: the overlap's premise)"""
  dtype = dtype or torch.float32
  t = de.HkvHashTable(torch.int64, dtype, torch.zeros(dim, dtype=dtype), init_capacity=cap, max_capacity=cap, device="cuda:0", dim=dim,
                      evict_strategy=de.HkvEvictStrategy.LRU, name=name)
  k = torch.from_numpy(fill_keys).cuda()
  for lo in range(0, k.numel(), 20000):
    kk = k[lo:lo + 20000]
    t._table.upsert(kk, (kk % 1000).to(torch.float32)[:, None].repeat(1, dim).to(dtype), unique_keys=True)
    torch.cuda.synchronize()
  for _ in range(3):   # the host learns the density from an asynchronous size read: give it calls to complete in
    t._table.upsert(k[:16], (k[:16] % 1000).to(torch.float32)[:, None].repeat(1, dim).to(dtype), unique_keys=True)
    torch.cuda.synchronize()
  return t


@pytest.mark.parametrize("cap,n", [(120_000, 5000), (1_200_000, 20000)])
def test_overlap_step_equals_sequential_ops_dictionary_oracle(env, cap, n):
  """Universe = 62 % of the slots (dense, yet both home buckets of a key are rarely full): next to nothing is evicted, so a
  dictionary is the oracle for every key that is present (the reference's CPU table semantics: last occurrence wins, misses read
  the default).  The small table has
  thousands of left-over keys per step (two keys of a batch sharing a home bucket), the large one a handful."""
  torch, de = env
  dim = 64
  rng = np.random.default_rng(cap)
  imin = np.iinfo(np.int64).min
  universe = rng.permutation(np.arange(1, int(cap * 0.62) + 1, dtype=np.int64)) * 7919 + 3
  resident = universe[: int(universe.size * 0.99)]          # the last 1 % enter through the steps (misses first)
  t = make_dense_table(torch, de, cap, dim, resident, "ovl_exact_%d" % cap)
  tbl = t._table
  latest = {int(k): float(int(k) % 1000) for k in resident}
  drv = de.OverlapAssignStep(t)
  nsteps = 10
  batches = []
  for s in range(nsteps + 1):
    ids = universe[(rng.zipf(1.15, size=n) * 37 + rng.integers(0, 50, size=n)) % universe.size].astype(np.int64)
    ids[rng.integers(0, n, size=n // 100)] = imin           # the two sentinel key values as ordinary keys
    ids[rng.integers(0, n, size=n // 130)] = imin + 1
    ids[: n // 6] = universe[7]                             # a hot id
    rng.shuffle(ids)
    batches.append(torch.from_numpy(ids).cuda())
  drv.prime(batches[0])
  n_evicted = 0
  for s in range(nsteps):
    ids = batches[s]
    vals = (torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim)
    look_ahead = batches[s + 1] if s % 4 != 3 else None      # every fourth step: the next ids are NOT announced
    ahead2 = batches[s + 2] if (s % 5 != 2 and look_ahead is not None and s + 2 <= nsteps) else None   # ... and not always two ahead
    out, ex = drv.step(vals, look_ahead, ahead2, return_exists=True)
    if look_ahead is None:
      drv.prime(batches[s + 1])
    ref, rex = tbl.find(ids, return_exists=True)            # the table right now = what this lookup had to reflect
    assert torch.equal(ex, rex), "step %d" % s
    assert torch.equal(out, ref), "step %d" % s
    ids_np = ids.cpu().numpy()
    want = np.array([latest.get(int(k), 0.0) for k in ids_np], np.float32)
    want_ex = np.array([int(k) in latest for k in ids_np])
    exn, outn = ex.cpu().numpy(), out[:, 0].cpu().numpy()
    # (a dense table DOES evict now and then — a key whose two home buckets are both full — so a key the dictionary holds may
    # be absent; never the other way round, and never more than a handful)
    assert not np.any(exn & ~want_ex)
    n_evicted += int(np.sum(~exn & want_ex))
    np.testing.assert_array_equal(outn[exn], want[exn])
    assert np.all(outn[~exn] == 0.0)
    assert bool((out == out[:, :1]).all())
    for i, k in enumerate(ids_np.tolist()):
      latest[k] = 100000.0 * (s + 1) + i
  assert n_evicted <= nsteps * n // 100, n_evicted
  drv.flush()
  st = drv.stats()
  assert st["overlapped"] == nsteps + 1 and st["sequential"] == 0 and not st["pending"], st   # (+1: the flush is a step without a lookup)
  ek, ev = t.export()
  assert ek.numel() == int(t.size().item()) <= len(latest) and ek.numel() >= 0.99 * len(latest)
  np.testing.assert_array_equal(ev[:, 0].cpu().numpy(), np.array([latest[int(k)] for k in ek.cpu().numpy()], np.float32))
  tbl.check_errors()
  assert tbl.slot_census()["locked"] == 0


def test_overlap_step_evictions_and_forced_conflicts(env):
  """A table filled to capacity, a third of every batch never-seen ids (each evicts the oldest entry of its two home buckets)
  and a third the OLDEST resident keys — the likely victims: the write-back keeps meeting victims the next lookup asks for.
  Checked against the table itself after every step (module docstring), plus: exists => the row of the key's last write;
  a key of the batch just written back is present; size <= capacity; the conflict path really ran."""
  torch, de = env
  dim, cap, n, nsteps = 64, 60_000, 3000, 40
  rng = np.random.default_rng(5)
  fill = np.arange(1, cap + 1, dtype=np.int64) * 104729 + 11
  t = make_dense_table(torch, de, cap, dim, fill, "ovl_evict")
  tbl = t._table
  last_write = {int(k): -1 for k in fill}       # key -> step of its last write (-1: the pre-fill); who is oldest
  latest = {int(k): float(int(k) % 1000) for k in fill}
  hot = fill[rng.integers(0, cap, size=400)]
  fresh = 10_000_000

  def make_batch(step):
    nonlocal fresh
    a = hot[(rng.zipf(1.2, size=n // 3)) % hot.size]
    b = np.arange(fresh, fresh + n // 3, dtype=np.int64) * 31 + 5
    fresh += n // 3
    oldest = sorted(last_write, key=last_write.get)[: 4 * n]
    c = np.array(oldest, np.int64)[rng.integers(0, len(oldest), size=n - a.size - b.size)]
    ids = np.concatenate([a, b, c])
    rng.shuffle(ids)
    return ids

  drv = de.OverlapAssignStep(t)
  ids_np = make_batch(0)
  drv.prime(torch.from_numpy(ids_np).cuda())
  prev_keys = None
  n_absent_old = 0
  keep_alive, nxt2_np, nxt2_t = [], None, None
  for s in range(nsteps):
    ids = torch.from_numpy(ids_np).cuda()
    vals = (torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim)
    for i, k in enumerate(ids_np.tolist()):     # (the next batch is drawn knowing this one's writes: who is oldest after it)
      last_write[k] = s
    # two batches are always announced — as the SAME device buffers from call to call, that is how the driver knows them again —
    # so the plans are built by the step launches
    if s == 0:
      nxt_np = make_batch(s + 1)
      nxt_t = torch.from_numpy(nxt_np).cuda()
    else:
      nxt_np, nxt_t = nxt2_np, nxt2_t
    nxt2_np = make_batch(s + 2)
    nxt2_t = torch.from_numpy(nxt2_np).cuda()
    keep_alive.append((ids, nxt_t, nxt2_t))
    out, ex = drv.step(vals, nxt_t, nxt2_t, return_exists=True)
    ref, rex = tbl.find(ids, return_exists=True)
    assert torch.equal(ex, rex), "step %d: %d exists flags differ" % (s, int((ex != rex).sum()))
    assert torch.equal(out, ref), "step %d" % s
    exn, outn = ex.cpu().numpy(), out[:, 0].cpu().numpy()
    want = np.array([latest.get(int(k), 0.0) for k in ids_np], np.float32)
    assert np.array_equal(outn[exn], want[exn]) and np.all(outn[~exn] == 0.0)
    if prev_keys is not None:
      again = np.isin(ids_np, prev_keys)
      assert again.any() and exn[again].all()   # written one step ago: present, served from that step's values
    n_absent_old += int((~exn).sum())
    for i, k in enumerate(ids_np.tolist()):
      latest[k] = 100000.0 * (s + 1) + i
    prev_keys = np.unique(ids_np)
    ids_np = nxt_np
    assert int(t.size().item()) <= tbl.capacity()
  drv.flush()
  st = drv.stats()
  # (the host learns that the table is dense from asynchronous size reads: the first steps may still run one op after the other)
  assert st["overlapped"] + st["sequential"] == nsteps + 1 and st["overlapped"] >= nsteps - 8, st
  assert st["deferred_evictions"] > 0 and st["victims_noted"] > 0 and st["rows_corrected"] > 0, st
  assert st["plans_built_in_launch"] >= nsteps - 10, st     # the two-launch, atomic-free plan build really ran
  ek, ev = t.export()
  ekn = ek.cpu().numpy()
  assert np.unique(ekn).size == ekn.size
  np.testing.assert_array_equal(ev[:, 0].cpu().numpy(), np.array([latest[int(k)] for k in ekn], np.float32))
  tbl.check_errors()
  assert tbl.slot_census()["locked"] == 0


def test_overlap_many_steps_one_host_call_and_fallback(env):
  """tfra_table_steps_overlap: 8 steps enqueued by ONE host call equal the same steps issued one by one; and a table the
  overlap does not cover (LFU scores: a key may be refused) runs the same entry points one op after the other."""
  torch, de = env
  dim, cap, n = 64, 200_000, 8192
  rng = np.random.default_rng(9)
  universe = np.arange(1, int(cap * 0.62) + 1, dtype=np.int64) * 6151 + 1
  tabs = [make_dense_table(torch, de, cap, dim, universe, "ovl_run_%d" % i) for i in range(2)]
  m = 8
  ids = [torch.from_numpy(universe[(rng.zipf(1.2, size=n) * 13) % universe.size]).cuda() for _ in range(m + 1)]
  vals = [(torch.randn((n, dim), device="cuda") * 0.01) for _ in range(m)]
  outs = [torch.empty((n, dim), device="cuda") for _ in range(m)]
  d0 = de.OverlapAssignStep(tabs[0])
  run = d0.make_run(ids[:m], vals, outs, ids_after=ids[m])
  run()
  d1 = de.OverlapAssignStep(tabs[1]).prime(ids[0])
  for k in range(m):
    o = d1.step(vals[k], ids[k + 1])
    # (twin tables: a dense table evicts now and then, and WHICH entry goes depends on the device clock — a handful of keys
    # may be resident in one table and not in the other; everything else must agree bit for bit)
    neq = (o != outs[k]).any(dim=1)
    assert int(neq.sum()) <= 8, (k, int(neq.sum()))
    assert bool(((o[neq] == 0).all(dim=1) | (outs[k][neq] == 0).all(dim=1)).all()), k
  d0.flush()
  d1.flush()
  assert d0.stats()["overlapped"] == m + 1 and d1.stats()["overlapped"] == m + 1
  a, b = tabs[0].export(), tabs[1].export()
  ka, kb = a[0].cpu().numpy(), b[0].cpu().numpy()
  common = np.intersect1d(ka, kb)
  assert common.size >= max(ka.size, kb.size) - 16
  ck = torch.from_numpy(common).cuda()
  assert torch.equal(tabs[0].lookup(ck), tabs[1].lookup(ck))
  # fallback: a table the overlap does not cover (LFU scores: a key may be refused) through the same driver
  g = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=1 << 20, max_capacity=1 << 22, device="cuda:0", dim=dim,
                      evict_strategy=de.HkvEvictStrategy.LFU, name="ovl_fallback")
  dg = de.OverlapAssignStep(g).prime(ids[0])
  ref = {}
  for k in range(4):
    o, ex = dg.step(vals[k], ids[k + 1], return_exists=True)
    idn = ids[k].cpu().numpy()
    want_ex = np.array([int(x) in ref for x in idn])
    np.testing.assert_array_equal(ex.cpu().numpy(), want_ex)
    for i, x in enumerate(idn.tolist()):
      ref[x] = (k, i)
  dg.flush()
  assert dg.stats()["sequential"] == 5 and dg.stats()["overlapped"] == 0
  uk = torch.from_numpy(np.array(sorted(ref), np.int64)).cuda()
  got = g.lookup(uk)
  want = torch.stack([vals[ref[int(x)][0]][ref[int(x)][1]] for x in uk.cpu().numpy()])
  assert torch.equal(got, want)


def _dict_check(torch, tbl, ids, out, ex, latest, tag):
  """rows / exists of one step against the table as it is now AND against the dictionary of last writes (keys that are present)"""
  ref, rex = tbl.find(ids, return_exists=True)
  assert torch.equal(ex, rex), tag
  assert torch.equal(out, ref), tag
  ids_np = ids.cpu().numpy()
  want = np.array([latest.get(int(k), 0.0) for k in ids_np], np.float32)
  want_ex = np.array([int(k) in latest for k in ids_np])
  exn, outn = ex.cpu().numpy(), out[:, 0].cpu().numpy()
  assert not np.any(exn & ~want_ex), tag
  np.testing.assert_array_equal(outn[exn], want[exn], err_msg=str(tag))
  assert np.all(outn[~exn] == 0.0), tag
  return int(np.sum(~exn & want_ex))


@pytest.mark.parametrize("pattern", ["phases", "restart8", "alternate"])
def test_overlap_plan_objects_see_regular_and_listless_builds_in_any_order(env, pattern):
  """The driver rotates four plan objects; each has two tables, and the two kinds of build leave them in different states (a
  regular build — a plan launch in front of the step — wants an empty target and empties the other table through its key list; a
  list-less build inside the step launch rewrites its target and empties nothing).  Regular, list-less, regular on ONE object used to
  start the third build in a table still holding the first batch's keys: stale keys were written back and forwarded.  Look-ahead that
  comes and goes (phases), a run of 8 steps without ids_after called again and again (restart8), every other step announced
  (alternate) — all against a dictionary of last writes."""
  torch, de = env
  dim, cap, n, nsteps = 64, 400_000, 6000, 40
  rng = np.random.default_rng({"phases": 1, "restart8": 2, "alternate": 3}[pattern])
  universe = rng.permutation(np.arange(1, int(cap * 0.62) + 1, dtype=np.int64)) * 7919 + 3
  t = make_dense_table(torch, de, cap, dim, universe, "ovl_mixed_%s" % pattern)
  tbl = t._table
  latest = {int(k): float(int(k) % 1000) for k in universe}
  batches = [torch.from_numpy(universe[(rng.zipf(1.15, size=n) * 37 + rng.integers(0, 50, size=n)) % universe.size].astype(np.int64)).cuda()
             for _ in range(nsteps + 3)]
  vals = [(torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim).contiguous() for s in range(nsteps)]
  drv = de.OverlapAssignStep(t)
  evicted = 0
  if pattern == "restart8":
    outs = [torch.empty((n, dim), device="cuda") for _ in range(8)]
    for lo in range(0, nsteps, 8):
      run = drv.make_run(batches[lo:lo + 8], vals[lo:lo + 8], outs, ids_after=None, values_before=None)
      run()
      drv.flush()
      torch.cuda.synchronize()
      # the table after the run = all 8 write-backs; the LAST lookup's rows are what the dictionary held before its own write
      for s in range(lo, lo + 8):
        if s == lo + 7:
          ids_np = batches[s].cpu().numpy()
          want = np.array([latest.get(int(k), 0.0) for k in ids_np], np.float32)
          got = outs[7][:, 0].cpu().numpy()
          present = got != 0.0
          np.testing.assert_array_equal(got[present], want[present])
        for i, k in enumerate(batches[s].cpu().numpy().tolist()):
          latest[k] = 100000.0 * (s + 1) + i
      ek = torch.from_numpy(np.array(sorted(latest), np.int64)).cuda()
      got, ex = t.lookup(ek, return_exists=True)
      exn = ex.cpu().numpy()
      assert exn.mean() > 0.99
      wantv = np.array([latest[int(k)] for k in ek.cpu().numpy()], np.float32)
      np.testing.assert_array_equal(got[:, 0].cpu().numpy()[exn], wantv[exn], err_msg="after run at %d" % lo)
  else:
    drv.prime(batches[0])
    for s in range(nsteps):
      if pattern == "phases":
        two = (s // 5) % 2 == 1          # five steps without any look-ahead, five with both batches announced, ...
        nxt = batches[s + 1] if two else None
        nx2 = batches[s + 2] if two else None
      else:
        nxt = batches[s + 1] if s % 3 != 2 else None     # batches 3k+2 are scattered at step 3k and built at step 3k+1 (list-less),
        nx2 = batches[s + 2] if s % 3 == 0 else None     # every other batch's plan is a launch of its own (regular)
      out, ex = drv.step(vals[s], nxt, nx2, return_exists=True)
      if nxt is None:
        drv.prime(batches[s + 1])
      evicted += _dict_check(torch, tbl, batches[s], out, ex, latest, (pattern, s))
      for i, k in enumerate(batches[s].cpu().numpy().tolist()):
        latest[k] = 100000.0 * (s + 1) + i
    drv.flush()
    st = drv.stats()
    assert st["plans_built_in_launch"] > 0 and st["plans_built_in_front"] > 0, st
  assert evicted <= nsteps * n // 100
  ek, ev = t.export()
  np.testing.assert_array_equal(ev[:, 0].cpu().numpy(), np.array([latest[int(k)] for k in ek.cpu().numpy()], np.float32))
  tbl.check_errors()
  assert tbl.slot_census()["locked"] == 0


def test_overlap_id_buffers_reused_at_the_same_address(env):
  """A ring of four id buffers rewritten in place once their batch has been written back, with look-ahead that is sometimes
  withdrawn: an announced batch is recognised by (address, length), so pairs scattered for a batch that is never stepped must not
  be taken for a later batch that lives at the same address."""
  torch, de = env
  dim, cap, n, nsteps = 64, 300_000, 5000, 36
  rng = np.random.default_rng(11)
  universe = rng.permutation(np.arange(1, int(cap * 0.62) + 1, dtype=np.int64)) * 104729 + 7
  t = make_dense_table(torch, de, cap, dim, universe, "ovl_ring")
  tbl = t._table
  latest = {int(k): float(int(k) % 1000) for k in universe}
  draw = lambda: torch.from_numpy(universe[(rng.zipf(1.15, size=n) * 41 + rng.integers(0, 64, size=n)) % universe.size].astype(np.int64)).cuda()
  ring = [torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(6)]
  vals = [(torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim).contiguous() for s in range(nsteps)]
  drv = de.OverlapAssignStep(t)
  # batch s lives in ring[s % 6]; the buffer is refilled only after batch s has been written back (two calls after its own)
  content = {}
  def fill(s):
    b = draw()
    ring[s % 6].copy_(b)
    content[s] = b.clone()
  for s in range(3):
    fill(s)
  drv.prime(ring[0])
  for s in range(nsteps):
    withdraw = s % 7 == 5            # announce s+2 now, then do NOT step it as announced: rewrite its buffer before its call
    out, ex = drv.step(vals[s], ring[(s + 1) % 6], ring[(s + 2) % 6], return_exists=True)
    torch.cuda.synchronize()
    _dict_check(torch, tbl, content[s], out, ex, latest, ("ring", s))
    for i, k in enumerate(content[s].cpu().numpy().tolist()):
      latest[k] = 100000.0 * (s + 1) + i
    fill(s + 3)
    if withdraw:
      # the input pipeline replaces batch s+2 (already scattered) — allowed only through a restart: flush, rewrite, prime
      drv.flush()
      fill(s + 1); fill(s + 2)
      drv.prime(ring[(s + 1) % 6])
  drv.flush()
  ek, ev = t.export()
  np.testing.assert_array_equal(ev[:, 0].cpu().numpy(), np.array([latest[int(k)] for k in ek.cpu().numpy()], np.float32))
  tbl.check_errors()


def test_overlap_step_without_prime_raises(env):
  torch, de = env
  t = make_dense_table(torch, de, 60_000, 64, np.arange(1, 40_000, dtype=np.int64), "ovl_noprime")
  drv = de.OverlapAssignStep(t)
  with pytest.raises(RuntimeError, match="prime"):
    drv.step(torch.zeros((4, 64), device="cuda"))


def test_overlap_all_distinct_ids(env):
  """Batches of all-distinct ids (uniform over a large universe): every second slot of the plan's table is taken, the driver cuts the
  write-back into smaller slices once it has seen a batch's distinct-key count (pinned memory) — same results either way."""
  torch, de = env
  dim, cap, n, nsteps = 64, 800_000, 8192, 12
  rng = np.random.default_rng(21)
  universe = rng.permutation(np.arange(1, int(cap * 0.62) + 1, dtype=np.int64)) * 7919 + 3
  t = make_dense_table(torch, de, cap, dim, universe, "ovl_distinct")
  tbl = t._table
  latest = {int(k): float(int(k) % 1000) for k in universe}
  batches = [torch.from_numpy(rng.choice(universe, size=n, replace=False)).cuda() for _ in range(nsteps + 2)]
  drv = de.OverlapAssignStep(t).prime(batches[0])
  for s in range(nsteps):
    vals = (torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim).contiguous()
    out, ex = drv.step(vals, batches[s + 1], batches[s + 2], return_exists=True)
    torch.cuda.synchronize()
    _dict_check(torch, tbl, batches[s], out, ex, latest, ("distinct", s))
    for i, k in enumerate(batches[s].cpu().numpy().tolist()):
      latest[k] = 100000.0 * (s + 1) + i
  drv.flush()
  st = drv.stats()
  assert st["overlapped"] >= nsteps - 4 and st["lookups_listed"] >= nsteps - 6, st
  tbl.check_errors()


@pytest.mark.parametrize("n", [1, 17, 63, 65, 1023, 1025])
def test_overlap_tiny_and_ragged_batches(env, n):
  """Batch sizes around the launch's granules (16 ids per lookup wave, 64 per block, 1024 per MAP segment / plan tile): one id, a
  partial wave, one over a block, one under / over a segment — with repeats, the two sentinel key values and changing look-ahead."""
  torch, de = env
  dim, cap, nsteps = 64, 200_000, 9
  rng = np.random.default_rng(100 + n)
  imin = np.iinfo(np.int64).min
  universe = rng.permutation(np.arange(1, int(cap * 0.62) + 1, dtype=np.int64)) * 7919 + 3
  t = make_dense_table(torch, de, cap, dim, universe, "ovl_tiny_%d" % n)
  tbl = t._table
  latest = {int(k): float(int(k) % 1000) for k in universe}

  def draw():
    ids = universe[rng.integers(0, 40, size=n)].copy()          # forty hot ids: every batch shares most of them with the one before
    if n > 2:
      ids[rng.integers(0, n)] = imin
      ids[rng.integers(0, n)] = imin + 1
    if n > 8:
      ids[rng.integers(0, n, size=n // 4)] = universe[rng.integers(0, universe.size, size=n // 4)]
    return torch.from_numpy(ids).cuda()

  batches = [draw() for _ in range(nsteps + 2)]
  drv = de.OverlapAssignStep(t).prime(batches[0])
  for s in range(nsteps):
    vals = (torch.arange(n, device="cuda", dtype=torch.float32) + 1000.0 * (s + 1))[:, None].repeat(1, dim).contiguous()
    nxt = batches[s + 1] if s % 4 != 3 else None
    nx2 = batches[s + 2] if (nxt is not None and s % 3 != 1) else None
    out, ex = drv.step(vals, nxt, nx2, return_exists=True)
    if nxt is None:
      drv.prime(batches[s + 1])
    torch.cuda.synchronize()
    _dict_check(torch, tbl, batches[s], out, ex, latest, ("tiny", n, s))
    for i, k in enumerate(batches[s].cpu().numpy().tolist()):
      latest[k] = 1000.0 * (s + 1) + i
  drv.flush()
  ek, ev = t.export()
  np.testing.assert_array_equal(ev[:, 0].cpu().numpy(), np.array([latest[int(k)] for k in ek.cpu().numpy()], np.float32))
  tbl.check_errors()


@pytest.mark.parametrize("ratio", [0.0, 0.5])
def test_assign_step_for_picks_a_driver_by_rule_and_both_give_the_sequential_results(env, ratio):
  """de.assign_step_for(table, new_key_ratio): the overlapped step up to a quarter never-seen ids per batch, the look-ahead driver
  beyond — one interface, the same results (lookup i+1 sees write-back i)."""
  torch, de = env
  assert de.assign_step_driver_for(0.0) == de.assign_step_driver_for(0.10) == "overlapped_step" and de.assign_step_driver_for(0.11) == "look_ahead"
  dim, cap, n, nsteps = 64, 300_000, 4000, 6
  rng = np.random.default_rng(31)
  universe = rng.permutation(np.arange(1, int(cap * 0.62) + 1, dtype=np.int64)) * 7919 + 3
  t = make_dense_table(torch, de, cap, dim, universe, "ovl_rule_%d" % int(ratio * 10))
  drv = de.assign_step_for(t, ratio)
  assert type(drv).__name__ == ("OverlapAssignStep" if ratio <= 0.25 else "_LookAheadAssignStep")
  latest = {int(k): float(int(k) % 1000) for k in universe}
  batches = [torch.from_numpy(universe[(rng.zipf(1.15, size=n) * 37) % universe.size]).cuda() for _ in range(nsteps + 2)]
  drv.prime(batches[0])
  for s in range(nsteps):
    vals = (torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim).contiguous()
    out, ex = drv.step(vals, batches[s + 1], batches[s + 2], return_exists=True)
    torch.cuda.synchronize()
    ids_np = batches[s].cpu().numpy()
    want = np.array([latest.get(int(k), 0.0) for k in ids_np], np.float32)
    exn = ex.cpu().numpy()
    np.testing.assert_array_equal(out[:, 0].cpu().numpy()[exn], want[exn])
    assert exn.mean() > 0.99
    for i, k in enumerate(ids_np.tolist()):
      latest[k] = 100000.0 * (s + 1) + i
  drv.flush()
  ek, ev = t.export()
  np.testing.assert_array_equal(ev[:, 0].cpu().numpy(), np.array([latest[int(k)] for k in ek.cpu().numpy()], np.float32))


def _foreign_hog(seconds, ready, stop):
  """second process: keeps cuda:0 busy with long kernels (large fp32 matrix products, queued deep) for `seconds`"""
  import time
  import torch
  a = torch.randn((6144, 6144), device="cuda")
  b = torch.randn((6144, 6144), device="cuda")
  torch.cuda.synchronize()
  ready.set()
  t0 = time.time()
  while time.time() - t0 < seconds and not stop.is_set():
    for _ in range(8):
      a = torch.mm(a, b) * 1e-4
    torch.cuda.synchronize()


@pytest.mark.parametrize("foreign", ["second_stream", "second_process"])
def test_overlap_steps_beside_foreign_work_on_the_gpu(env, foreign):
  """The step launch's tail blocks wait (bounded) for the write-back blocks of the SAME launch; the launch assumes that its blocks
  get onto the chip in grid order.  Foreign work — another stream of this process queued deep with long kernels, or ANOTHER PROCESS
  doing the same — takes CUs away and delays residency: 400 overlapped steps next to it must still equal the ops run one after the
  other, with no bounded wait timing out (tfra_table_check_errors counts every one) and nothing left locked.
  Reference semantics: K/hkv_hashtable_op_gpu.cu.cc:192-213,256-267 (Insert exclusive, Find shared)."""
  torch, de = env
  import threading
  dim, cap, n, nsteps = 64, 2_000_000, 20000, 400
  rng = np.random.default_rng(21)
  universe = rng.permutation(np.arange(1, int(cap * 0.9) + 1, dtype=np.int64)) * 7919 + 3
  t = make_dense_table(torch, de, cap, dim, universe[: int(universe.size * 0.95)], "ovl_foreign_" + foreign)
  tbl = t._table
  batches = []
  for s in range(nsteps + 2):
    ids = universe[(rng.zipf(1.15, size=n) * 37 + rng.integers(0, 5000, size=n)) % universe.size].astype(np.int64)
    batches.append(torch.from_numpy(ids).cuda())
  vals = [(torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (k + 1))[:, None].repeat(1, dim) for k in range(4)]
  stop_flag = threading.Event()
  proc = None
  if foreign == "second_stream":
    side = torch.cuda.Stream()
    a = torch.randn((6144, 6144), device="cuda")
    b = torch.randn((6144, 6144), device="cuda")

    def hog():
      with torch.cuda.stream(side):
        x = a
        while not stop_flag.is_set():
          for _ in range(8):
            x = torch.mm(x, b) * 1e-4
          side.synchronize()
    th = threading.Thread(target=hog, daemon=True)
    th.start()
  else:
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ready, stop = ctx.Event(), ctx.Event()
    proc = ctx.Process(target=_foreign_hog, args=(60.0, ready, stop))
    proc.start()
    assert ready.wait(120), "the foreign process did not come up"
  try:
    drv = de.OverlapAssignStep(t).prime(batches[0])
    mism = 0
    for s in range(nsteps):
      out, ex = drv.step(vals[s % 4], batches[s + 1], batches[s + 2], return_exists=True)
      if s % 8 == 7:    # (every step would serialise the two workloads at the host: the foreign queue must stay deep while steps run)
        ref, rex = tbl.find(batches[s], return_exists=True)
        mism += int((ex != rex).sum()) + int((out != ref).any(dim=1).sum())
    drv.flush()
    torch.cuda.synchronize()
  finally:
    stop_flag.set()
    if proc is not None:
      stop.set()
      proc.join(90)
      if proc.is_alive():
        proc.kill()
    else:
      th.join(60)
  assert mism == 0, mism
  st = drv.stats()
  assert st["overlapped"] >= nsteps - 8, st
  tbl.check_errors()                       # a tail block that gave up waiting would have counted an error
  assert tbl.slot_census()["locked"] == 0
  # the last batch's writes are all there (nothing dropped)
  last = batches[nsteps - 1]
  got, ex = tbl.find(last, return_exists=True)
  lastv = vals[(nsteps - 1) % 4]
  uk, inv = torch.unique(last, return_inverse=True)
  lp = torch.zeros(uk.numel(), dtype=torch.long, device="cuda")
  lp.scatter_reduce_(0, inv, torch.arange(n, device="cuda"), reduce="amax", include_self=False)
  assert bool(ex.all()) and torch.equal(got, lastv[lp][inv])


def test_overlap_step_on_an_epoch_lru_table(env):
  """Round 6: EPOCHLRU tables take the overlapped launch too (score = epoch << 32 | clock: like LRU a new key is always admitted —
  the premise of the forwarding; T/hkv_hashtable_evict_test.py:481-525 for the strategy).  Every step equals a plain find of the
  table right after the call; every launch was the overlapped one; and the EPOCH a key's score carries is the epoch of its last
  write, exactly as on a twin table driven one op after the other (same step_per_epoch: the epoch advances per write-back)."""
  torch, de = env
  dim, cap, n, nsteps, spe = 64, 200_000, 6000, 26, 3
  rng = np.random.default_rng(77)
  universe = rng.permutation(np.arange(1, int(cap * 0.62) + 1, dtype=np.int64)) * 6151 + 1

  def make(name):
    t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=cap, max_capacity=cap, device="cuda:0", dim=dim,
                        evict_strategy=de.HkvEvictStrategy.EPOCHLRU, step_per_epoch=spe, name=name)
    k = torch.from_numpy(universe).cuda()
    for lo in range(0, k.numel(), 20000):      # (the same number of upserts on both tables: the same epoch when the steps start)
      kk = k[lo:lo + 20000]
      t._table.upsert(kk, (kk % 1000).to(torch.float32)[:, None].repeat(1, dim), unique_keys=True)
      torch.cuda.synchronize()
    for _ in range(3):
      t._table.upsert(k[:16], (k[:16] % 1000).to(torch.float32)[:, None].repeat(1, dim), unique_keys=True)
      torch.cuda.synchronize()
    return t

  ta, tb = make("ovl_elru_a"), make("ovl_elru_b")
  batches = [torch.from_numpy(universe[(rng.zipf(1.15, size=n) * 37 + rng.integers(0, 50, size=n)) % universe.size]).cuda() for _ in range(nsteps + 2)]
  drv = de.OverlapAssignStep(ta).prime(batches[0])
  for s in range(nsteps):
    vals = (torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim)
    out, ex = drv.step(vals, batches[s + 1], batches[s + 2], return_exists=True)
    ref, rex = ta._table.find(batches[s], return_exists=True)
    assert torch.equal(ex, rex) and torch.equal(out, ref), "step %d" % s
    tb._table.find(batches[s])                       # the twin: one op after the other
    tb._table.upsert_sparse(batches[s], vals)
  drv.flush()
  st = drv.stats()
  assert st["overlapped"] >= nsteps - 8 and st["why_sequential"] in (0, 32), st     # (32: not yet known dense — the first steps only)
  ka, _, sa = ta.export_with_scores(1)
  kb, _, sb = tb.export_with_scores(1)
  da = dict(zip(ka.cpu().numpy().tolist(), (sa.cpu().numpy().astype(np.uint64) >> np.uint64(32)).tolist()))
  db = dict(zip(kb.cpu().numpy().tolist(), (sb.cpu().numpy().astype(np.uint64) >> np.uint64(32)).tolist()))
  both = [k for k in da if k in db]
  assert len(both) >= 0.99 * max(len(da), len(db))
  assert all(da[k] == db[k] for k in both)
  assert max(da.values()) >= nsteps // spe           # the epoch really advanced while the steps ran
  ta._table.check_errors()
  assert ta._table.slot_census()["locked"] == 0


@pytest.mark.parametrize("new_share", [0.02, 0.3])
def test_overlap_step_on_a_growing_table_dictionary_exact(env, new_share):
  """Round 6: a GROWING table (CuckooHashTable: no eviction, no max_capacity — TFRA's default creator) takes the overlapped launch:
  every key of a batch ends up in the table, so the forwarding premise holds; new keys that find no free slot at home go to the tail's
  walking path; the table GROWS in front of a launch when it must (here several times: it starts at 8192 slots' worth).  Nothing is
  ever evicted, so a dictionary is the exact oracle (the reference's CPU cuckoo table semantics: K/cuckoo_hashtable_op.cc:111-150,
  last occurrence wins; lookup i+1 sees update i): every step's rows AND exists flags, and the final export, key for key."""
  torch, de = env
  dim, n, nsteps = 64, 20000, 40
  rng = np.random.default_rng(int(new_share * 100))
  t = de.CuckooHashTable(torch.int64, torch.float32, torch.full((dim,), -1.0), device="cuda:0", dim=dim, name="ovl_grow_%d" % int(new_share * 100))
  tbl = t._table
  resident = rng.permutation(np.arange(1, 60001, dtype=np.int64)) * 7919 + 3
  k = torch.from_numpy(resident).cuda()
  tbl.upsert(k, (k % 1000).to(torch.float32)[:, None].repeat(1, dim), unique_keys=True)
  latest = {int(x): float(int(x) % 1000) for x in resident}
  fresh = 10_000_000
  imin = np.iinfo(np.int64).min
  batches = []
  for s in range(nsteps + 2):
    ids = resident[(rng.zipf(1.15, size=n) * 37 + rng.integers(0, 50, size=n)) % resident.size].astype(np.int64)
    m = int(n * new_share)
    ids[rng.choice(n, size=m, replace=False)] = np.arange(fresh, fresh + m, dtype=np.int64) * 31 + 5     # never-seen keys
    fresh += m
    ids[rng.integers(0, n, size=3)] = imin
    ids[rng.integers(0, n, size=3)] = imin + 1
    batches.append(torch.from_numpy(ids).cuda())
  cap0 = tbl.capacity()
  drv = de.OverlapAssignStep(t).prime(batches[0])
  for s in range(nsteps):
    ids = batches[s]
    vals = (torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim)
    nxt = batches[s + 1] if s % 7 != 6 else None
    nx2 = batches[s + 2] if (nxt is not None and s % 5 != 4) else None
    out, ex = drv.step(vals, nxt, nx2, return_exists=True)
    if nxt is None:
      drv.prime(batches[s + 1])
    ids_np = ids.cpu().numpy()
    want = np.array([latest.get(int(x), -1.0) for x in ids_np], np.float32)
    want_ex = np.array([int(x) in latest for x in ids_np])
    np.testing.assert_array_equal(ex.cpu().numpy(), want_ex, err_msg="step %d" % s)
    np.testing.assert_array_equal(out[:, 0].cpu().numpy(), want, err_msg="step %d" % s)
    assert bool((out == out[:, :1]).all())
    for i, x in enumerate(ids_np.tolist()):
      latest[x] = 100000.0 * (s + 1) + i
  drv.flush()
  st = drv.stats()
  assert st["overlapped"] >= nsteps and st["sequential"] <= 1, st
  assert int(t.size().item()) == len(latest)
  if new_share > 0.1:
    assert tbl.capacity() > cap0                                  # it really grew while the steps ran
  ek, ev = t.export()
  ekn = ek.cpu().numpy()
  assert np.unique(ekn).size == ekn.size == len(latest)
  np.testing.assert_array_equal(ev[:, 0].cpu().numpy(), np.array([latest[int(x)] for x in ekn], np.float32))
  assert bool((ev == ev[:, :1]).all())
  tbl.check_errors()


def test_overlap_step_on_a_bounded_table_that_never_fills(env):
  """An Hkv LRU table whose max_capacity is far away (a common deployment) grows like the cuckoo flavour and never evicts: the
  overlapped launch from the first step on, a dictionary the exact oracle, growth while the steps run."""
  torch, de = env
  dim, n, nsteps = 64, 12000, 30
  rng = np.random.default_rng(8)
  t = de.HkvHashTable(torch.int64, torch.float32, torch.full((dim,), -1.0), init_capacity=1 << 16, max_capacity=1 << 24, device="cuda:0", dim=dim,
                      evict_strategy=de.HkvEvictStrategy.LRU, name="ovl_never_full")
  tbl = t._table
  resident = rng.permutation(np.arange(1, 20001, dtype=np.int64)) * 7919 + 3
  k = torch.from_numpy(resident).cuda()
  tbl.upsert(k, (k % 1000).to(torch.float32)[:, None].repeat(1, dim), unique_keys=True)
  latest = {int(x): float(int(x) % 1000) for x in resident}
  fresh = 5_000_000
  batches = []
  for s in range(nsteps + 2):
    ids = resident[(rng.zipf(1.15, size=n) * 37 + rng.integers(0, 50, size=n)) % resident.size].astype(np.int64)
    m = n // 2                                           # 6 K never-seen keys per step: past the first growth mark after ~20 steps
    ids[rng.choice(n, size=m, replace=False)] = np.arange(fresh, fresh + m, dtype=np.int64) * 31 + 5
    fresh += m
    batches.append(torch.from_numpy(ids).cuda())
  cap0 = tbl.capacity()
  drv = de.OverlapAssignStep(t).prime(batches[0])
  for s in range(nsteps):
    vals = (torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim)
    out, ex = drv.step(vals, batches[s + 1], batches[s + 2], return_exists=True)
    ids_np = batches[s].cpu().numpy()
    np.testing.assert_array_equal(ex.cpu().numpy(), np.array([int(x) in latest for x in ids_np]), err_msg="step %d" % s)
    np.testing.assert_array_equal(out[:, 0].cpu().numpy(), np.array([latest.get(int(x), -1.0) for x in ids_np], np.float32), err_msg="step %d" % s)
    for i, x in enumerate(ids_np.tolist()):
      latest[x] = 100000.0 * (s + 1) + i
  drv.flush()
  st = drv.stats()
  assert st["overlapped"] >= nsteps and st["sequential"] <= 1, st
  assert int(t.size().item()) == len(latest) and tbl.capacity() > cap0
  ek, ev = t.export()
  ekn = ek.cpu().numpy()
  assert np.unique(ekn).size == ekn.size == len(latest)
  np.testing.assert_array_equal(ev[:, 0].cpu().numpy(), np.array([latest[int(x)] for x in ekn], np.float32))
  tbl.check_errors()


def test_overlap_step_while_a_bounded_table_fills_up_and_starts_evicting(env):
  """From a third full to beyond capacity through the SAME driver: growing phase, at max_capacity but sparse (new keys walk), dense
  (evictions in the launch, deferred evictions, corrections).  Every step equals a plain find of the table right after the call; the
  size never passes the capacity; nothing locked, no error."""
  torch, de = env
  dim, n, nsteps, cap = 64, 8000, 60, 1 << 17
  rng = np.random.default_rng(12)
  t = de.HkvHashTable(torch.int64, torch.float32, torch.zeros(dim), init_capacity=cap // 2, max_capacity=cap, device="cuda:0", dim=dim,
                      evict_strategy=de.HkvEvictStrategy.LRU, name="ovl_fills_up")
  tbl = t._table
  resident = rng.permutation(np.arange(1, cap // 3 + 1, dtype=np.int64)) * 7919 + 3
  k = torch.from_numpy(resident).cuda()
  tbl.upsert(k, (k % 1000).to(torch.float32)[:, None].repeat(1, dim), unique_keys=True)
  fresh = 9_000_000
  batches = []
  for s in range(nsteps + 2):
    ids = resident[(rng.zipf(1.15, size=n) * 37 + rng.integers(0, 50, size=n)) % resident.size].astype(np.int64)
    m = n // 3                                           # 2.7 K never-seen keys per step: the table is full after ~35 steps
    ids[rng.choice(n, size=m, replace=False)] = np.arange(fresh, fresh + m, dtype=np.int64) * 31 + 5
    fresh += m
    batches.append(torch.from_numpy(ids).cuda())
  drv = de.OverlapAssignStep(t).prime(batches[0])
  for s in range(nsteps):
    vals = (torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * (s + 1))[:, None].repeat(1, dim)
    out, ex = drv.step(vals, batches[s + 1], batches[s + 2], return_exists=True)
    ref, rex = tbl.find(batches[s], return_exists=True)
    assert torch.equal(ex, rex) and torch.equal(out, ref), "step %d" % s
    assert int(t.size().item()) <= tbl.capacity()
  drv.flush()
  st = drv.stats()
  assert st["overlapped"] >= nsteps - 2, st
  assert int(t.size().item()) > 0.9 * tbl.capacity()      # it did fill up
  assert st["deferred_evictions"] + st["victims_noted"] >= 0
  tbl.check_errors()
  assert tbl.slot_census()["locked"] == 0
  # the last batch's writes are there
  last = batches[nsteps - 1]
  got, ex = tbl.find(last, return_exists=True)
  lastv = (torch.arange(n, device="cuda", dtype=torch.float32) + 100000.0 * nsteps)[:, None].repeat(1, dim)
  uk, inv = torch.unique(last, return_inverse=True)
  lp = torch.zeros(uk.numel(), dtype=torch.long, device="cuda")
  lp.scatter_reduce_(0, inv, torch.arange(n, device="cuda"), reduce="amax", include_self=False)
  assert bool(ex.all()) and torch.equal(got, lastv[lp][inv])
