"""Build libmmdm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmdm_hip.so")
SOURCES = ["mmdm.hip", "gemm_f32.hip", "gemm_bf16.hip", "gemm_split.hip", "attn_f32.hip", "rowops.hip", "geometry.hip"]


# Translation units compiled WITHOUT the packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  Measured in round 5
# (tools/canary.hip, tools/overlap_bisect.py; LAB_NOTES.md): on gfx950 a packed-fp32 instruction (seen: the low lane of v_pk_mul_f32 ... op_sel_hi:[1,0])
# transiently delivers a WRONG result when its wave shares a SIMD with waves of the packed-W GEMM kernels (gemm_splitw / gemm_bf16w); wait
# states behind it (up to 8, hand-assembled) change nothing -- pure
# register arithmetic of an unrelated kernel gives other bits, and the rotation round trip of the geometry kernels amplifies one such bit
# into a turned joint.  Without these instructions nothing moves (0 of 1e10 evaluations).  The geometry kernels run once per step: the flag
# costs nothing.  (rowops.hip -- AdaLN, 640 launches per step beside the other stream's GEMMs -- was built this way too for one profile round:
# +1.3 ms per fp32 step; its kernels were never seen to move -- the denoiser outputs were bit-equal in every wrong step of the hunt, and
# tools/adaln_victim.py holds 0 of its launches moving beside the packed GEMMs -- so it keeps the packed instructions.)
NO_PACKED_FP32 = {"geometry.hip"}
NO_PACKED_FP32_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


SHA_FILES = {"fp32": ("gemm_f32.hip", "mmdm.hip", "kernels.h"), "fp32_split": ("gemm_split.hip", "mmdm.hip", "kernels.h"),
             "bf16": ("gemm_bf16.hip", "mmdm.hip", "kernels.h"), "bf16_fp8": ("gemm_bf16.hip", "mmdm.hip", "kernels.h")}


def sources_sha(precision="fp32"):
    """sha256 over the sources of the kernel a committed PMC traffic figure belongs to -- the dominant GEMM of the precision mode
    (csrc/gemm_f32.hip / gemm_split.hip / gemm_bf16.hip), the host orchestration csrc/mmdm.hip that decides which GEMMs a step launches,
    and csrc/kernels.h: profiles/gemm_traffic*.json record it so that bench.py can tell whether that measurement describes the kernels
    it is running."""
    import hashlib
    h = hashlib.sha256()
    for f in SHA_FILES[precision]:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith(".o")]      # sources, kernels.h, the linker map
    deps.append(os.path.join(HERE, "..", "include", "mmdm.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, "kernels.h"), os.path.join(HERE, "..", "include", "mmdm.h")]
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in [path] + hdrs):
            continue                                  # object is newer than its source and the shared headers
        # -fvisibility=hidden: the shared library exports exactly what include/mmdm.h declares (its declarations sit inside a visibility pragma)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-c", path, "-o", obj]
        if src in NO_PACKED_FP32:
            cmd[5:5] = NO_PACKED_FP32_FLAGS
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))    # translation units are independent: compile them side by side
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "libmmdm.map"), "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
