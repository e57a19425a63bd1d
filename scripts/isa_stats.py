"""Static instruction mix of the gfx950 kernels of one .hip source (no GPU needed): compiles the device side to
assembly and counts, per kernel, VALU / SALU / VMEM / LDS / waitcnt / branch instructions and the register use.
   python scripts/isa_stats.py recommenders-addons_amd/csrc/tfra_csr.hip [name-substring ...]"""
import collections, os, re, subprocess, sys, tempfile

src = sys.argv[1]
pats = sys.argv[2:]
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only",
                       "-S", src, "-o", out], stderr=subprocess.DEVNULL)
cur, stats, regs, last_name = None, collections.OrderedDict(), {}, None
for line in open(out):
  m = re.match(r"^(_Z\w+):", line)
  if m:
    cur = m.group(1); stats[cur] = collections.Counter(); continue
  if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
    cur = None
  m = re.match(r"\s+\.name:\s+(\S+)", line)
  if m: last_name = m.group(1)
  m = re.match(r"\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\d+)", line)
  if m and last_name: regs.setdefault(last_name, {})[m.group(1)] = int(m.group(2))
  if cur is None: continue
  t = line.strip().split()
  if not t or t[0].startswith((".", ";")) or t[0].endswith(":"): continue
  op = t[0]
  c = stats[cur]
  if op.startswith("s_waitcnt"): c["wait"] += 1
  elif op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
  elif op.startswith("s_"): c["salu"] += 1
  elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
    c["vmem"] += 1
    if "atomic" in op: c["atomic"] += 1
  elif op.startswith("ds_"): c["lds"] += 1
  elif op.startswith("v_"):
    c["valu"] += 1
    if "mul" in op or "mad_u64" in op: c["vmul"] += 1
    if "readlane" in op or "readfirstlane" in op: c["rdlane"] += 1
  else: c["other"] += 1
for k, c in stats.items():
  d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
  if pats and not any(p in d for p in pats): continue
  short = re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "").replace("void ", ""))
  print("%-50s valu %4d (mul %3d) salu %4d branch %3d vmem %3d (atomic %2d) lds %3d wait %3d | %s" % (
      short[:50], c["valu"], c["vmul"], c["salu"], c["branch"], c["vmem"], c["atomic"], c["lds"], c["wait"], regs.get(k, {})))
