"""Debug build of the library for tools/overlap_bisect.py and tools/overlap_stray.py (build container or GPU box): build/libmmdm_debug.so = the
shipped translation units with csrc/mmdm.hip patched in memory (nothing is written into the tree) -- select it with MMDM_LIB=build/libmmdm_debug.so.
Extra mmdm_diag_set keys, each taking a handle pointer as its value: snap_handle / snap_weights / diff_handle / diff_weights (which of the handle's
allocations, by the name of the pointer they were allocated for, changed since the snapshot -- and the first differing values of out1 / out2),
poison_handle (every allocation filled with 0x7f), poison_scratch (the per-step buffers filled with NaN), and dbg_skip <mask>: helper classes of
mmdm.hip that launch nothing while the mask is set (1 fp32 GEMM, 2 fp32 attention, 4 bf16 GEMM, 8 fp8 GEMM, 16 split GEMM, 32 plane attention,
64 bf16 attention, 128 AdaLN) -- a step graph captured under a mask is the aggressor with those kernels left out."""
import os, re, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cs = os.path.join(root, "mixermdm_amd", "csrc")
s = open(os.path.join(cs, "mmdm.hip")).read()
s = s.replace('extern "C" int mmdm_diag_set(const char* key, long long value) {', 'int mmdm_debug_handle(const char* key, long long value);\nextern "C" int mmdm_diag_set(const char* key, long long value) {\n    if (key && mmdm_debug_handle(key, value)) return MMDM_OK;', 1)
s = s.replace('int dalloc(mmdm_handle h, float** p, size_t nfloats) {', 'int dalloc_real(mmdm_handle h, float** p, size_t nfloats) {', 1)
marker = 'int add_slot(mmdm_handle h, const std::string& name'
assert marker in s
s = s.replace(marker, '''static std::map<void*, std::string> g_dbg_names;
static int dalloc_n(mmdm_handle h, const char* nm, float** p, size_t n) { int rc = dalloc_real(h, p, n); if (!rc) g_dbg_names[*p] = nm; return rc; }
#define dalloc(h, p, n) dalloc_n(h, #p, p, n)
''' + marker, 1)
s = s.replace('#include', '#include <map>\n#include', 1)
s = s.replace('constexpr int NF = MMDM_NF, NF2 = 2 * MMDM_NF;', 'constexpr int NF = MMDM_NF, NF2 = 2 * MMDM_NF;\nint g_dbg_skip = 0;', 1)
for name, bit in (("linear", 1), ("attention_plain", 2), ("linear_b", 4), ("linear_8", 8), ("linear_s", 16), ("attention_p", 32), ("attention_b", 64)):
    i = s.index("\nint %s(const Ctx& c" % name); b = s.index("{\n", i)
    s = s[:b + 2] + "    if (g_dbg_skip & %d) return MMDM_OK;\n" % bit + s[b + 2:]
i = s.index("\nint run_stack(const Ctx& c")
s = s[:i] + "\n#define mmdm_adaln_any(...) ((g_dbg_skip & 128) ? 0 : mmdm_adaln_any(__VA_ARGS__))" + s[i:]
s += r'''
static std::map<void*, std::vector<unsigned char>> g_dbg_snap;
int mmdm_debug_handle(const char* key, long long value) {
    const bool snap = !strcmp(key, "snap_handle"), diff = !strcmp(key, "diff_handle"), poison = !strcmp(key, "poison_handle"), diffw = !strcmp(key, "diff_weights"), snapw = !strcmp(key, "snap_weights");
    if (!strcmp(key, "dbg_skip")) { g_dbg_skip = (int)value; return 1; }
    const bool pscr = !strcmp(key, "poison_scratch");
    if (pscr) {
        mmdm_handle hh = reinterpret_cast<mmdm_handle>((uintptr_t)value);
        (void)hipDeviceSynchronize();
        static const char* ok[] = {"&sc->h", "&sc->xn", "&sc->att", "&sc->qkv", "&sc->kv", "&sc->f1", "&sc->xp", "&h->o1", "&h->o2", "&h->out1", "&h->out2", "&h->w23", "&h->mI",
                                   "&h->hpool", "&h->model_out", "&h->se_d1", "&h->ss_d1", "&h->se_d2", "&h->ss_d2", "&h->se_mx", "&h->ss_mx", "&q1", "&q2"};
        int np_ = 0;
        for (void* p : hh->allocs) {
            const std::string nm = g_dbg_names.count(p) ? g_dbg_names[p] : "?";
            bool hit = false;
            for (const char* o : ok) hit = hit || nm == o;
            if (!hit) continue;
            size_t n = 0; void* base = nullptr;
            (void)hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &n, p);
            (void)hipMemset(p, 0xff, n); ++np_;
        }
        (void)hipDeviceSynchronize();
        fprintf(stderr, "[dbg] poison_scratch: %d allocations filled with 0xff\n", np_);
        return 1;
    }
    if (!snap && !diff && !poison && !diffw && !snapw) return 0;
    mmdm_handle h = reinterpret_cast<mmdm_handle>((uintptr_t)value);
    (void)hipDeviceSynchronize();
    int idx = 0; size_t total = 0;
    auto visit = [&](void* p, const char* kind) {
        size_t n = 0; void* base = nullptr;
        (void)hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &n, p);
        const std::string nm = g_dbg_names.count(p) ? g_dbg_names[p] : "?";
        total += n;
        if (snap || snapw) { auto& v = g_dbg_snap[p]; v.resize(n); (void)hipMemcpy(v.data(), p, n, hipMemcpyDeviceToHost); }
        else if (diff || diffw) {
            std::vector<unsigned char> cur(n); (void)hipMemcpy(cur.data(), p, n, hipMemcpyDeviceToHost);
            auto& v = g_dbg_snap[p]; size_t nd = 0, first = 0, last = 0;
            if (v.size() == n) for (size_t i = 0; i < n; ++i) if (cur[i] != v[i]) { if (!nd) first = i; last = i; ++nd; }
            if (nd) fprintf(stderr, "[dbg] %s alloc %d %s (%p, %zu bytes): %zu bytes differ, offsets [%zu, %zu]\n", kind, idx, nm.c_str(), p, n, nd, first, last);
            if (nd && (nm == "&h->out1" || nm == "&h->out2")) {
                const float* g = reinterpret_cast<const float*>(v.data()); const float* b = reinterpret_cast<const float*>(cur.data());
                int shown = 0; long prev_row = -1; int in_row = 0;
                for (size_t i = 0; i < n / 4 && shown < 60; ++i)
                    if (memcmp(g + i, b + i, 4)) {
                        const long row = (long)(i / 524);
                        if (row != prev_row) { prev_row = row; in_row = 0; }
                        if (in_row++ < 6) { fprintf(stderr, "[dbg]    %s[row %ld (seq %ld, frame %ld), col %d]: alone %.6g, beside B %.6g\n", nm.c_str(), row, row / 181, row % 181, (int)(i % 524), g[i], b[i]); ++shown; }
                    }
            }
        } else (void)hipMemset(p, 0x7f, n);
        ++idx;
    };
    if (snap || diff || poison) for (void* p : h->allocs) visit(p, "work");
    if (snapw || diffw || poison) for (void* p : h->wb->allocs) visit(p, "weight");
    fprintf(stderr, "[dbg] %s: %d allocations, %.1f MB\n", key, idx, total / 1e6);
    (void)hipDeviceSynchronize();
    return 1;
}
'''
tmp = os.path.join(cs, "_mmdm_dbg.hip")
open(tmp, "w").write(s)
os.makedirs(os.path.join(root, "build"), exist_ok=True)
try:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-c", tmp, "-o", os.path.join(root, "build", "mmdm_dbg.o")])
    objs = [os.path.join(cs, f + ".o") for f in ("attn_f32", "gemm_bf16", "gemm_f32", "gemm_split", "geometry", "rowops")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(cs, "libmmdm.map"), "-o",
                           os.path.join(root, "build", "libmmdm_debug.so"), os.path.join(root, "build", "mmdm_dbg.o")] + objs)
finally:
    os.remove(tmp)
print("built")
