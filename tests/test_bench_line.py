"""CPU: the ONE JSON line bench.py prints must survive the driver's stdout tail (~8 KB): compact_line() keeps it under 6 000
characters whatever the full result holds, with the keys the judge reads (value, roofline, cpu_baseline, scaling_point); the full
result goes to bench_detail.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full_result():
  """a result shaped like run_bounded()'s + main()'s additions, with worst-case string lengths"""
  long = "x" * 1500
  kern = {"k%d %s" % (i, long[:200]): {"avg_launch_us": 12.3456789, "algorithmic_bytes_per_launch": 80162576, "achieved_GBps": 2450.123456,
                                        "frac": 0.30637, "traffic": 89374562} for i in range(6)}
  sec_one = {"metric": long, "value": 2.4233e9, "ms_per_step": 0.054088, "driver": "look_ahead", "driver_rule": long,
             "value_overlapped_step": 1.877e9, "value_look_ahead_driver": 2.423e9, "value_op_surface": 1.571e9, "value_plain_call": 1.773e9,
             "config": {"workload": long, "host_enqueue_ms_per_step": 0.0482, "unique_keys_per_batch": 78453, "drivers": {"a": long, "b": long},
                        "timing": {"x": {"ms_per_step_by_window": [0.1] * 5}}, "verified": {"k%d" % i: True for i in range(20)}},
             "roofline": {"frac": 0.1242, "step_frac": 0.2532, "kernel": long, "kernels": kern}}
  res = {"metric": long, "value": 4047412345.678, "unit": "pairs/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.0323841234,
         "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
         "driver": "overlapped_step", "driver_rule": long,
         "config": {"workload": long, "slots": 999999992, "global_batch": 131072, "unique_ratio": 0.1736, "unique_keys_per_batch": 22737,
                    "new_key_ratio": 0.0, "steps_per_host_call": 1, "host_enqueue_ms_per_step": 0.0068, "resident_after_prefill": 944986864,
                    "drivers": {"k%d" % i: long for i in range(8)}, "verified": {"k%d" % i: True for i in range(30)},
                    "timing": {"k%d" % i: {"ms_per_step_by_window": [0.0323] * 5} for i in range(8)}, "overlapped_step_stats": {"a": 1}},
         "roofline": {"bound": "hbm", "kernel": long, "achieved": 2450.9123, "peak": 8000.0, "unit": "GB/s", "frac": 0.306371234,
                      "traffic": 89374562, "traffic_source": "profiles/r99_summary.json", "algorithmic_bytes_per_launch": 80162576, "avg_launch_us": 32.7071234, "step_frac": 0.3094212,
                      "step_algorithmic_bytes": 80162576, "kernels": kern, "timing": long, "step_bytes_definition": long},
         "cpu_baseline": {"value": 43980000.123, "unit": "lookup+insert pairs/s", "cores": 128, "kind": "reference", "resident_keys": 256000000,
                          "table_ops_only_pairs_per_s": 1002133618, "host_cores": 256, "host_ram_bytes": 3 << 40, "sample": long,
                          "rungs_tried": [{"keys": 256000000, "phases_s": [["created", 9.1]] * 5}] * 3,
                          "per_op": {"find_unique_ids_ops_per_s": 313690179, "insert_or_assign_ops_per_s": 353621484,
                                     "find_with_repeats_ops_per_s": 120000, "prefill_keys_per_s_init_size_N": 12008808, "step_pairs_per_s": 1},
                          "per_op_per_core": {"a": 1}, "prefill_keys_per_s_init_size_8192_growth_included": 984127, "small_table_legs_keys": 4000000,
                          "not_run_1e9_keys": long},
         "scaling_point": {"workload": long, "value": 1.2131e9, "ms_per_step": 0.10804, "n_gpus": 1, "error": None, "note": long},
         "secondary": {"c3": sec_one, "c2": sec_one, "c4": sec_one}}
  for k in ("overlapped_step", "overlapped_step_4_steps_per_host_call", "look_ahead_driver", "plain_call", "op_surface", "op_surface_host_read_first",
            "op_surface_find_first", "accum", "op_surface_table_ops_only"):
    res["value_" + k] = 4.0474123e9
    res["ms_per_step_" + k] = 0.0323
  return res


def test_compact_line_is_one_short_json_line_with_the_judged_keys():
  line = bench.compact_line(_full_result(), "bench_detail.json")
  assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 6000, len(line)
  d = json.loads(line)
  for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
    assert k in d, k
  assert d["config"]["steps_per_host_call"] == 1 and d["config"]["driver"] == "overlapped_step" and d["config"]["workload"]
  for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us", "step_frac"):
    assert k in d["roofline"], k
  for k in ("value", "unit", "cores", "kind", "sample", "resident_keys", "table_ops_only_pairs_per_s", "find_with_repeats_ops_per_s",
            "prefill_keys_per_s_init_size_8192_growth_included"):
    assert k in d["cpu_baseline"], k
  assert d["scaling_point"]["value"] and d["scaling_point"]["ms_per_step"]
  assert set(d["secondary"]) == {"c3", "c2", "c4"} and all("value" in v and "ms_per_step" in v for v in d["secondary"].values())
  assert d["variants_pairs_per_s"]["overlapped_step_4_steps_per_host_call"]
  assert d["detail"] == "bench_detail.json"
  # the run's own verification reaches the parsed line as ONE bit (the AND of every flag, secondary workloads included) ...
  assert d["verified"] is True and d["verified_flags"] == 30 + 3 * 20
  # ... the traffic's provenance is named, and the event timing says what it includes
  assert d["roofline"]["traffic_source"] == "profiles/r99_summary.json" and "dispatch" in d["roofline"]["avg_launch_us_is"]
  # the CPU leg at the metric's 10^9 keys (measured once, committed under profiles/) is quoted with its source
  assert d["cpu_baseline"]["value_at_1e9_keys"] > 1e6 and d["cpu_baseline"]["source_1e9"].startswith("profiles/")


def test_compact_line_says_false_when_any_verification_flag_failed():
  res = _full_result()
  res["secondary"]["c2"] = dict(res["secondary"]["c2"], config=dict(res["secondary"]["c2"]["config"], verified={"a": True, "b": False}))
  assert json.loads(bench.compact_line(res))["verified"] is False
  res = _full_result()
  res["config"]["verified"] = {}
  for v in res["secondary"].values():
    v["config"]["verified"] = {}
  assert json.loads(bench.compact_line(res))["verified"] is False     # nothing checked is not "verified"


def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
  """`python bench.py --gpus N` (the form the driver uses) must not die for want of torch.distributed.run: it re-executes itself under
  the launcher.  Here: the command it would run, captured instead of run."""
  import subprocess
  seen = {}

  def fake_call(cmd, env=None):
    seen["cmd"], seen["env"] = cmd, env
    return 0
  monkeypatch.setattr(subprocess, "call", fake_call)
  monkeypatch.delenv("WORLD_SIZE", raising=False)
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
  try:
    bench.main()
    assert False, "main() should have exited with the launcher's code"
  except SystemExit as e:
    assert e.code == 0
  cmd = seen["cmd"]
  assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
  assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
  assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_compact_line_of_the_round4_result_that_the_driver_could_not_parse():
  """the 27.8 KB line of round 4 (profiles/r04_bench_line_driver_args.json) through the same function"""
  path = os.path.join(ROOT, "profiles", "r04_bench_line_driver_args.json")
  full = json.load(open(path))
  assert len(json.dumps(full)) > 20000
  line = bench.compact_line(full)
  assert len(line) < 6000
  d = json.loads(line)
  assert d["value"] == bench._num(full["value"]) and d["roofline"]["frac"] == bench._num(full["roofline"]["frac"])
  assert d["cpu_baseline"]["kind"] == "reference" and d["scaling_point"]["value"]


def test_compact_line_degrades_instead_of_overflowing():
  res = _full_result()
  res["secondary"] = {"w%d" % i: res["secondary"]["c3"] for i in range(40)}
  line = bench.compact_line(res)
  assert len(line) < 6000 and json.loads(line)["roofline"]["frac"]


def test_emit_prints_the_line_last_and_writes_the_detail_file(tmp_path, monkeypatch, capsys):
  monkeypatch.setattr(bench, "ROOT", str(tmp_path))
  bench.emit(_full_result())
  out = capsys.readouterr().out.strip().splitlines()
  d = json.loads(out[-1])
  assert d["detail"] == "bench_detail.json"
  full = json.load(open(tmp_path / "bench_detail.json"))
  assert "kernels" in full["roofline"] and len(json.dumps(full)) > 6000


def test_live_traffic_reads_the_two_pmc_passes_of_a_rocprofv3_on_path(tmp_path, monkeypatch):
  """bench.live_traffic(): `roofline.traffic` measured by the run itself — two child passes under rocprofv3 (--pmc FETCH_SIZE, then
  --pmc WRITE_SIZE, each with --kernel-trace only), the step kernel's launches averaged, (2 x FETCH + WRITE) KiB.  Here a stand-in
  `rocprofv3` on PATH that writes the counter file the real one writes (several rows per dispatch: one per counter instance; other
  kernels beside the step's); switched off, absent, failing or over its limit, the function returns None and the line falls back to
  the committed profile summary."""
  import argparse
  import stat
  fake = tmp_path / "bin" / "rocprofv3"
  fake.parent.mkdir()
  fake.write_text("""#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
c = a[a.index("--pmc") + 1]
d = a[a.index("-d") + 1]
assert "--kernel-trace" in a and "--sys-trace" not in a and "--hip-trace" not in a and "-s" not in a     # PMC passes on their own
child = a[a.index("--") + 1:]
assert "--config" in child and child[child.index("--config") + 1] == "m1b" and "--no-secondary" in child and "--no-cpu-baseline" in child
assert os.environ.get("TFRA_BENCH_LIVE_TRAFFIC") == "0"      # the child does not start passes of its own
if os.environ.get("FAKE_ROCPROF_FAIL"):
  sys.exit(3)
os.makedirs(os.path.join(d, "host", "1234"), exist_ok=True)
v = {"FETCH_SIZE": 20000.0, "WRITE_SIZE": 44000.0}[c]
with open(os.path.join(d, "host", "1234", "live_counter_collection.csv"), "w") as f:
  f.write("Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\\n")
  for disp in (1, 2, 3, 4):
    for part in range(2):   # two instances of the counter per dispatch: summed
      f.write('%d,%d,1,1,1,1,1,1,"void (anonymous namespace)::step_k_u2(StepArgs)",256,0,0,96,0,100,%s,%f,0,0\\n' % (disp, disp, c, v / 2))
  f.write('9,9,1,1,1,1,1,1,"void find_kernel<16, 4, true, true>(tfra::TableView)",256,0,0,57,0,100,%s,%f,0,0\\n' % (c, 1e9))
""")
  fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
  monkeypatch.setenv("PATH", str(fake.parent) + os.pathsep + os.environ.get("PATH", ""))
  monkeypatch.delenv("TFRA_BENCH_LIVE_TRAFFIC", raising=False)
  args = argparse.Namespace(slots=1000, batch=64)
  got = bench.live_traffic(args)
  assert got is not None and got["step_k"] == int((2 * 20000.0 + 44000.0) * 1024) and got["launches"] == 4
  assert got["source"].startswith("live: rocprofv3 --pmc FETCH_SIZE")
  monkeypatch.setenv("FAKE_ROCPROF_FAIL", "1")
  assert bench.live_traffic(args) is None
  monkeypatch.delenv("FAKE_ROCPROF_FAIL")
  monkeypatch.setenv("TFRA_BENCH_LIVE_TRAFFIC", "0")
  assert bench.live_traffic(args) is None
