"""Registers, spills, LDS and scratch of the kernels in a code object: reads the AMDGPU metadata notes
of a built object file (its .hip_fatbin section unbundled, llvm-readelf --notes on the gfx950 code object).
Usage: python profiles/kregs.py <file.o> [name-substring ...]"""
import re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1]
pats = sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    co, fat = os.path.join(d, "co"), os.path.join(d, "fat")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, text=True).stdout
cur = {}
rows = []
for ln in notes.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip().strip("'")
    if k == "agpr_count" and cur:
        rows.append(cur); cur = {}
    cur[k] = v
if cur:
    rows.append(cur)
for r in rows:
    name = r.get("name", "")
    if "vgpr_count" not in r:
        continue
    if pats and not any(p in name for p in pats):
        continue
    dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
    print("%-110s vgpr %3s agpr %3s sgpr %3s spill %3s lds %6s scratch %5s" % (
        dem[:110], r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"), r.get("vgpr_spill_count"),
        r.get("group_segment_fixed_size"), r.get("private_segment_fixed_size")))
