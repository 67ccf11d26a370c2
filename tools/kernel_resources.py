"""Registers, spills, scratch and LDS of every kernel in librsba.so (from the code object's metadata notes).
usage: python tools/kernel_resources.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("RSBA_LIB") or os.path.join(ROOT, "realsensecalibration_amd", "librsba.so")
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    co = os.path.join(d, "rsba.co")
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + os.path.join(d, "fat"),
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co] if False else ["true"])
    # the fat binary sits in section .hip_fatbin of the shared object
    fat = os.path.join(d, "fat")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", LIB, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    syms = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--symbols", "--wide", co], text=True)
size = {}   # code bytes of every kernel (the FUNC symbol's size)
for ln in syms.splitlines():
    f = ln.split()
    if len(f) >= 8 and f[3] == "FUNC":
        size[f[7]] = int(f[2])
rows = []
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    rows.append((name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), blk.split()[0], size.get(g("name"), 0)))
pat = sys.argv[1:]
print("%-90s %5s %6s %6s %8s %8s %5s %8s" % ("kernel", "vgpr", "vspill", "sspill", "scratch", "lds", "agpr", "code B"))
for r in sorted(rows):
    if not pat or any(p in r[0] for p in pat):
        print("%-90s %5s %6s %6s %8s %8s %5s %8d" % (r[0][:90], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
