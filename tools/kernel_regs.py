"""Register / LDS / spill counts of the kernels in a compiled translation unit (CPU side, no GPU): python tools/kernel_regs.py mixermdm_amd/csrc/gemm_f32.o [filter]"""
import re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin/"
obj, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as d:
    fat, dev = os.path.join(d, "fat.bin"), os.path.join(d, "dev.o")
    subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dev])
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", dev], capture_output=True, text=True).stdout
ks = re.split(r"\n\s+- \.agpr_count", notes)[1:]
rows = []
for k in ks:
    k = ".agpr_count" + k
    g = lambda key: (re.search(re.escape(key) + r":\s+(\S+)", k) or [None, "?"])[1]
    rows.append((g(".name"), g(".vgpr_count"), g(".agpr_count"), g(".vgpr_spill_count"), g(".sgpr_count"), g(".group_segment_fixed_size")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("vgpr(total) agpr spill sgpr lds  name")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if flt in n:
        print(f"{r[1]:>5s} {r[2]:>4s} {r[3]:>4s} {r[4]:>4s} {r[5]:>6s}  {n[:150]}")
