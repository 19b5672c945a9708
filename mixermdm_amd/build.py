"""Build libmmdm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmdm_hip.so")

# Translation units compiled WITHOUT the packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  Measured in round 5
# (tools/canary.hip, tools/overlap_bisect.py; LAB_NOTES.md): on gfx950 the rotation round trip of the geometry kernels -- dense VALU code with
# packed-fp32 instructions in it -- transiently computes OTHER bits while its wave shares a SIMD with waves of the packed-W GEMM kernels
# (gemm_splitw / gemm_bf16w); hand-assembled wait states behind every packed instruction change nothing, and the same code without packed
# instructions never moves (0 of 1e10 evaluations).  The root cause (which ingredient of the packed-W kernels it takes) is OPEN, so the rule is
# by construction, not by observation: every kernel that can run beside a packed-W GEMM of this library -- i.e. every kernel of a
# precision 1-3 handle, and the geometry kernels of any handle (another handle's packed GEMMs may be on the device) -- is built without
# packed-fp32 instructions, unless a test pins its bits beside the aggressor (DESIGN.md section 4 lists the remaining sites and their guards):
#   geometry.hip        every handle: no packed-fp32 instructions (its kernels run once per step: no cost)
#   gemm_bf16.hip, gemm_fp8p.hip, gemm_split.hip   the GEMMs of precision 1-3 handles (and the aggressors themselves): their epilogues held ~36 000 such
#                       sites; without them the steps are FASTER (A/B on one box, round 6: fp32_split 26.02 -> 25.67 ms/step, bf16_fp8 9.29 -> 9.20 --
#                       the guide's "packed f32 VALU beside MFMAs is an anti-lever"), so the flag costs less than nothing
#   rowops.hip          built TWICE: rowops.o (packed; precision 0 handles and the stateless entry points: no aggressor beside them unless the
#                       caller brings one, include/mmdm.h) and rowops_nopk.o (-DMMDM_ROWOPS_NOPK: the same kernels under *_nopk names, taken
#                       by precision 1-3 handles: AdaLN, LayerNorm, cond SiLU, time mean, Influence head, MDM pack/unpack)
# The feature flag is a cc1 option and reaches the x86 host pass too, which answers "'-packed-fp32-ops' is not a recognized feature for this
# target (ignoring feature)" five times per file; -Xarch_device cannot carry an option that takes an argument (clang refuses), and the
# function-level form (#pragma clang attribute ... target("no-packed-fp32-ops")) stops the device library's functions from being inlined.
# The build therefore filters exactly that line from the compiler's stderr.
NO_PACKED_FP32_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
_HOST_PASS_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)"

# (source, object, extra flags)
UNITS = [
    ("mmdm.hip", "mmdm.o", []),
    ("gemm_f32.hip", "gemm_f32.o", []),
    ("gemm_bf16.hip", "gemm_bf16.o", NO_PACKED_FP32_FLAGS),
    ("gemm_fp8p.hip", "gemm_fp8p.o", NO_PACKED_FP32_FLAGS),
    ("gemm_split.hip", "gemm_split.o", NO_PACKED_FP32_FLAGS),
    ("attn_f32.hip", "attn_f32.o", []),
    ("rowops.hip", "rowops.o", []),
    ("rowops.hip", "rowops_nopk.o", ["-DMMDM_ROWOPS_NOPK"] + NO_PACKED_FP32_FLAGS),
    ("geometry.hip", "geometry.o", NO_PACKED_FP32_FLAGS),
]
SOURCES = sorted({u[0] for u in UNITS})
NO_PACKED_FP32_OBJECTS = [u[1] for u in UNITS if "-packed-fp32-ops" in u[2]]
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]


SHA_FILES = {"fp32": ("gemm_f32.hip", "mmdm.hip", "kernels.h"), "fp32_split": ("gemm_split.hip", "mmdm.hip", "kernels.h"),
             "bf16": ("gemm_bf16.hip", "mmdm.hip", "kernels.h"), "bf16_fp8": ("gemm_bf16.hip", "gemm_fp8p.hip", "gemm_fp8p.h", "mmdm.hip", "kernels.h")}

_COMMENT_OR_STRING = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\\n])*"|\'(?:\\.|[^\'\\\n])*\'', re.S)


def strip_comments(text):
    """C / C++ source without its comments (string and character literals are kept as they are), blank lines and trailing blanks dropped:
    what sources_sha() hashes, so that a comment can be corrected without voiding the profiles tied to the code."""
    def repl(m):
        s = m.group(0)
        return s if s[0] in "\"'" else (" " if s.startswith("/*") else "")
    out = _COMMENT_OR_STRING.sub(repl, text)
    return "\n".join(ln.rstrip() for ln in out.split("\n") if ln.strip())


def sources_sha(precision="fp32"):
    """sha256 over the CODE (comments stripped: strip_comments) of the kernel a committed PMC traffic figure belongs to -- the dominant GEMM of
    the precision mode (csrc/gemm_f32.hip / gemm_split.hip / gemm_bf16.hip), the host orchestration csrc/mmdm.hip that decides which GEMMs a step
    launches, and csrc/kernels.h: profiles/gemm_traffic*.json record it so that bench.py can tell whether that measurement describes the
    kernels it is running."""
    h = hashlib.sha256()
    for f in SHA_FILES[precision]:
        h.update(f.encode())
        h.update(strip_comments(open(os.path.join(CSRC, f), encoding="utf-8").read()).encode())
    return h.hexdigest()[:16]


def _cmd(hipcc, src, obj, extra):
    return [hipcc] + BASE_FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", os.path.join(CSRC, obj)]


def _cmd_stamp(cmd):
    """What an object was built with: the command line (paths relative to the tree) -- stored beside the object as <obj>.cmd, compared on every build,
    so that an object built with other flags (an older checkout, an edited flag set, a maintainer's own recipe) is never reused."""
    return " ".join(os.path.relpath(a, HERE) if a.startswith(HERE) else a for a in cmd)


def _object_stale(cmd, src, obj, hdrs):
    objp, stamp = os.path.join(CSRC, obj), os.path.join(CSRC, obj + ".cmd")
    if not os.path.exists(objp) or not os.path.exists(stamp):
        return True
    if open(stamp).read() != _cmd_stamp(cmd):
        return True
    return any(os.path.getmtime(d) > os.path.getmtime(objp) for d in [os.path.join(CSRC, src)] + hdrs)


_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _headers(src=None):
    """Headers an object depends on: the quoted includes of its source, followed recursively (csrc/*.h, include/mmdm.h)."""
    if src is None:
        return sorted({h for s_, _, _ in UNITS for h in _headers(s_)})
    seen, todo = set(), [os.path.join(CSRC, src)]
    while todo:
        f = todo.pop()
        for inc in _INCLUDE.findall(open(f, encoding="utf-8").read()):
            h = os.path.normpath(os.path.join(os.path.dirname(f), inc))
            if os.path.exists(h) and h not in seen:
                seen.add(h)
                todo.append(h)
    return sorted(seen)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".map"))]      # sources, kernels.h, the linker map
    deps.append(os.path.join(HERE, "..", "include", "mmdm.h"))
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return any(_object_stale(_cmd(hipcc, s, o, x), s, o, _headers(s)) for s, o, x in UNITS)


def build(force=False, verbose=True, jobs=None):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = jobs or int(os.environ.get("MMDM_BUILD_JOBS", "8"))
    pending = []
    for src, obj, extra in UNITS:
        # -fvisibility=hidden: the shared library exports exactly what include/mmdm.h declares (its declarations sit inside a visibility pragma)
        cmd = _cmd(hipcc, src, obj, extra)
        if force or _object_stale(cmd, src, obj, _headers(src)):
            pending.append((cmd, obj))
    running = []

    def reap(block_all):
        while running and (block_all or len(running) >= jobs):
            cmd, obj, pr = running.pop(0)
            _, err = pr.communicate()
            err = "\n".join(ln for ln in err.decode(errors="replace").split("\n") if _HOST_PASS_NOISE not in ln)
            if err.strip():
                sys.stderr.write(err if err.endswith("\n") else err + "\n")
            if pr.returncode != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
            with open(os.path.join(CSRC, obj + ".cmd"), "w") as f:
                f.write(_cmd_stamp(cmd))
    for cmd, obj in pending:                          # translation units are independent: compile them side by side
        reap(False)
        if verbose:
            print(" ".join(cmd), flush=True)
        stamp = os.path.join(CSRC, obj + ".cmd")
        if os.path.exists(stamp):
            os.remove(stamp)                          # (a failed or interrupted compile leaves no stamp behind)
        running.append((cmd, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE)))
    reap(True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "libmmdm.map"), "-o", LIB] + \
          [os.path.join(CSRC, o) for _, o, _ in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
