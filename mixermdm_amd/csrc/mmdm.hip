// Host side of libmmdm_hip.so: handle, packed weights, workspace, per-step orchestration and hipGraph replay.
//
// One step = MixerDiffusion.ddim_sample on the CFG-doubled batch (reference call stack: SURVEY.md 3.2;
// src/models/utils/gaussian_diffusion.py:1871-1965, src/models/utils/cfg_sampler.py:38-56, src/models/mixermdm.py:660-810).
// Everything the reference re-uploads per step (timestep tensors, schedule scalars, normaliser stats) lives on the
// device; the step index is a device word that the kernels read, so ONE captured graph serves every step.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "kernels.h"

// ------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int mmdm_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int mmdm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return MMDM_OK;
}

int mmdm_kernels_init(void) {
    static int state = -1;  // -1 not tried, 0 ok
    if (state == 0) return MMDM_OK;
    int rc = mmdm_gemm_init();
    if (!rc) rc = mmdm_gemm_bf16_init();
    if (!rc) rc = mmdm_gemm_split_init();
    if (!rc) rc = mmdm_attn_init();
    if (!rc) state = 0;
    return rc;
}

// Several handles may be driven from several host threads (mmdm_create_shared).  This runtime survives concurrent graph LAUNCHES, but not a
// capture / instantiation / exec destruction beside another thread's capture or launch (segfaults inside hipGraphLaunch, seen with two
// threads at the real model sizes): captures, instantiations and evictions take this lock exclusively, replays take it shared.
static std::shared_mutex g_graph_mu;
// Sampling calls of different handles OVERLAP on the device in every precision mode (MMDM_SERIALIZE_HANDLES=1 serialises them again: below).
// For most of round 5 the calls of low-precision handles were serialised: two such handles side by side gave wrong motions.  The cause was
// not in this file: on gfx950 the geometry kernels' rotation round trip -- dense VALU code that held packed-fp32 instructions (v_pk_mul_f32 /
// v_pk_fma_f32 / v_pk_add_f32) -- transiently computes OTHER bits while its wave shares a SIMD with waves of the packed-W GEMM kernels
// (gemm_splitw / gemm_bf16w), and turned one such bit into a turned joint.  The result is wrong, not late (hand-assembled wait states behind
// every packed instruction change nothing), the same code without packed-fp32 instructions never moves, and WHICH ingredient of the packed-W
// kernels it takes is not isolated (LAB_NOTES.md; tools/canary.hip, tools/overlap_bisect.py).  The rule is therefore by construction
// (mixermdm_amd/build.py): geometry.hip is built without packed-fp32 instructions for every handle, and precision 1-3 handles take the
// row kernels (AdaLN, LayerNorm, cond SiLU, time mean, MDM pack / unpack) of rowops_nopk.o -- the second build of rowops.hip, without them --
// through ROWOP below; tests/test_gpu_ragged.py keeps full-size handles of every precision pair side by side bit-exact.
#define ROWOP(c, fn, ...) (((c).h && (c).h->cfg.precision != 0) ? fn##_nopk(__VA_ARGS__) : fn(__VA_ARGS__))
// MMDM_SERIALIZE_HANDLES=1 (read once, by the first mmdm_create): every sampling call (mmdm_run) waits on the device for the previous
// sampling call of ANY handle of the process -- one process-wide event, no host synchronisation -- so that overlap between handles can be
// ruled out in the field as the cause of a wrong motion.  Off by default.
static std::atomic<int> g_serialize_handles{-1};
static std::mutex g_serial_mu;
static hipEvent_t g_serial_ev = nullptr;        // recorded behind the last sampling call of any handle (g_serial_mu)

static thread_local char g_gemm_note[160] = "";
void mmdm_note_gemm_reset(void) { g_gemm_note[0] = 0; }
void mmdm_note_gemm(const char* fmt, ...) {
    size_t n = strlen(g_gemm_note);
    if (n && n + 1 < sizeof(g_gemm_note)) g_gemm_note[n++] = '+';
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_gemm_note + n, sizeof(g_gemm_note) - n, fmt, ap);
    va_end(ap);
}
extern "C" const char* mmdm_last_gemm_kernel(void) { return g_gemm_note; }

extern "C" const char* mmdm_last_error(void) { return g_err; }

// The ONE diagnostic entry point (include/mmdm.h section 4): process-global switches of the scripts under tools/ and of bench.py's
// in-loop clock measurement.  Each key belongs to one translation unit; nothing here is read by a handle's default launches except
// through the diagnostic kernel instantiations the switches select.
extern "C" int mmdm_diag_set(const char* key, long long value) {
    if (!key) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_diag_set: null key");
    if (mmdm_diag_gemm_f32(key, value) || mmdm_diag_gemm_bf16(key, value) || mmdm_diag_gemm_split(key, value) || mmdm_diag_attn(key, value)) return MMDM_OK;
    return mmdm_set_error(MMDM_ERR_ARG, "mmdm_diag_set: unknown key \"%s\"", key);
}
extern "C" const char* mmdm_version(void) { return "gfx950;mmdm-hip r6"; }

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));    \
    } while (0)
#define RC(expr)                     \
    do {                             \
        int _rc = (expr);            \
        if (_rc) return _rc;         \
    } while (0)

namespace {

constexpr int NF = MMDM_NF, NF2 = 2 * MMDM_NF;
constexpr int NFS = 320;  // pose width of the embedding GEMM on the fp32-split kernel (precision >= 1): K a multiple of its 64-element packed step
constexpr int NFP = 272;  // pose width padded to the GEMM's K step (16 floats): motion_embed weights are stored [D, 272] with zero columns and
                          // the embedding GEMMs read repacked, zero-padded pose rows (mmdm_repack_pose), so they run on the LDS-DMA kernel

// ------------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------------
struct Slot {            // one destination of mmdm_set_weight
    float* dst = nullptr;
    int64_t rows = 0, cols = 0;   // expected source shape
    int64_t ld = 0;               // destination row stride (>= cols)
    bool required = true, set = false;
};

struct LayerW {
    float *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b;
    float *ca_in_w, *ca_in_b, *ca_out_w, *ca_out_b;
    float *f1_w, *f1_b, *f2_w, *f2_b;
};

struct LayerWB {         // bf16 twins of the stack GEMM weights (precision >= 1), converted at mmdm_prepare
    void *sa_in_w = nullptr, *sa_out_w = nullptr, *ca_in_w = nullptr, *ca_out_w = nullptr, *f1_w = nullptr, *f2_w = nullptr;
    // precision == 3: the QKV / cross-attention input projections and both FFN matrices as fp8 e4m3 + one scale per output channel
    void *sa_in_8 = nullptr, *ca_in_8 = nullptr, *f1_8 = nullptr, *f2_8 = nullptr;
    float *sa_in_s = nullptr, *ca_in_s = nullptr, *f1_s = nullptr, *f2_s = nullptr;
};

struct StackW {          // a transformer stack: denoiser blocks or Influence blocks
    int D = 0, F = 0, L = 0, H = 0, n_ada = 0;
    bool has_ca = false;
    bool w_packed = false;                      // low-precision weight twins stored in MFMA fragment order (gemm_splitw_kernel / gemm_bf16w_kernel take W straight from global memory)
    float *ada_w = nullptr, *ada_b = nullptr;   // [L*n_ada*2D, D], [L*n_ada*2D]  (slots: sa, [ca_q, ca_kv,] ffn)
    void* ada_s = nullptr;                      // precision >= 1: ada_w as two fp16 planes in fragment order (cond_vectors on the fp32-split kernel)
    std::vector<LayerW> layers;
    std::vector<LayerWB> layers_b;
};

struct EncLayerW {       // one nn.TransformerEncoderLayer (post-norm): MDMDenoiser.seqTransEncoder.layers.{i}  src/models/mdm.py:252-264
    float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_g, *n1_b, *n2_g, *n2_b;
};

struct ModuleW {         // denoiser or mixer front/back ends
    StackW st;
    int kind = 0;                                // 0: AdaLN blocks (in2IN / InterGen / Influence); 1: MDMDenoiser (post-norm encoder + cond token)
    std::vector<EncLayerW> enc;                  // kind == 1
    float *pe = nullptr;                         // [5000, D]
    float *me_w = nullptr, *me_b = nullptr;      // motion_embed [D, 262] stored with ld NFP
    void* me_s = nullptr;                        // precision >= 1: motion_embed as two fp16 planes [2][D][NFS] in fragment order (embedding on the fp32-split kernel)
    float *te_w = nullptr, *te_b = nullptr;      // text_embed [D, text_dim]
    float *t0_w = nullptr, *t0_b = nullptr, *t2_w = nullptr, *t2_b = nullptr;   // embed_timestep.time_embed.{0,2}
    float *out_w = nullptr, *out_b = nullptr;    // out.linear [262, D] / influence.out [nw, D]
    float *time_tab = nullptr;                   // [S, D] = time_embed(pe[timestep_map])  (built by set_schedule)
    float *pe_r = nullptr;                       // ragged call: [rows, D] = pe[frame index of every row of a group] (gathered by mmdm_begin_ragged; this handle's own)
};

struct Scratch {          // transformer-stack work buffers (one set per concurrently running stack)
    float *h = nullptr, *xn = nullptr, *qkv = nullptr, *kv = nullptr, *att = nullptr, *f1 = nullptr;
    float *xp = nullptr;      // [2][rows][NFP] repacked pose operands of the embedding GEMMs (fp32), or [2][2 planes][rows][NFS] fp16
    void *qk = nullptr, *kvp = nullptr;   // precision >= 1: bf16 plane copies of the attention's Q|K ([planes][R][2D]) and cross-attention K ([planes][R][D])
    float* xs = nullptr;                  // precision == 3: per-row scales of the fp8 AdaLN output in xn
};

// Row geometry of the current sampling call.  Uniform: B items x T frames, sequence s = rows [s T, (s + 1) T).  Ragged (mmdm_begin_ragged): the B
// items' frames lie back to back in a GROUP of `rows` rows (sum of the lengths, rounded up to the handle's row bucket); a buffer of k B
// sequences is k groups; where a sequence starts, how long it is and which sequence / frame a row belongs to are DEVICE arrays, so a captured
// step graph depends on (B, rows, query tiles of the longest item) only.
struct Geom {
    int B = 0, T = 0;                // items; frames (ragged: the longest item -- sizes the attention grid only)
    bool rag = false;
    int rows = 0;                    // frame rows of one group of B sequences (uniform: B * T)
    int real_rows = 0;               // ragged: sum of the lengths (rows - real_rows padding rows per group)
    double tt1 = 0;                  // sum over the items of T_i (T_i + 1): attention FLOP accounting
    const int *seq_off = nullptr, *seq_len = nullptr, *row_seq = nullptr, *item_order = nullptr;
    mmdm_rag rg{nullptr, nullptr, nullptr, nullptr, 0, 0};
    size_t rows_of(int nseq) const { return rag ? (size_t)(nseq / B) * rows : (size_t)nseq * T; }
};

struct Prof {
    bool on = false;
    static constexpr int NCLS = 4;               // 0: GEMMs in the handle's own operand type (fp32 / bf16 / split planes), 1: attention, 2: fp8-operand GEMMs
                                                 // (precision 3), 3: fp32 GEMMs of a low-precision handle (embeddings, conditioning, heads: their roof is the fp32 one)
    std::vector<hipEvent_t> ev[NCLS];            // pairs (start, stop) per launch, per class
    size_t used[NCLS] = {0, 0, 0, 0};
    double flops[NCLS] = {0, 0, 0, 0};
    double bytes[NCLS] = {0, 0, 0, 0};           // algorithmic bytes (operands read once + result written once)
};

}  // namespace

// Device memory of a handle's WEIGHTS (fp32 parameters, packed AdaLN matrices, the low-precision twins made by mmdm_prepare, normaliser
// statistics): owned by this block, which the creating handle and every handle made from it by mmdm_create_shared hold by reference -- K
// sampler handles (own workspace, streams, schedule tables and graph cache each) over ONE copy of the 1.46 GB parameter set.  Freed when the
// last holder is destroyed, on the device it was allocated on.
struct mmdm_weight_block {
    std::vector<void*> allocs;
    int device = 0;
    ~mmdm_weight_block() {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
};

struct mmdm_handle_s {
    mmdm_config cfg;
    char err[512] = "";
    std::vector<void*> allocs;                       // workspace, schedule tables, step state: this handle's own
    std::shared_ptr<mmdm_weight_block> wb;           // weights: shared with the handles created from this one (mmdm_create_shared)
    bool alloc_weights = false;                      // dalloc() target while the weight slots are being laid out
    bool shared_child = false;                       // created by mmdm_create_shared: weights are set / prepared through the parent only
    std::unordered_map<std::string, Slot> slots;
    ModuleW d1, d2, mx;
    int nw = 23;
    bool prepared = false;

    // schedule
    int S = 0;
    int* d_tmap = nullptr;       // [Smax]
    float* d_coef = nullptr;     // [4*Smax]
    float* d_stats = nullptr;    // [4*262]
    bool stats_set = false;
    int* d_step = nullptr;       // [2]: step_idx, loop_pos
    int host_step = -1;          // mirror of step_idx
    int h_tmap0 = 0;             // host copy of timestep_map[0] (mmdm_module_forward borrows slot 0 of the tables and puts it back)
    int Smax = 1000;

    // call state
    int B = 0, T = 0;
    bool begun = false;
    Geom geom;                                             // row geometry of the begun call (uniform or ragged)
    int rag_bucket = 128;                                  // ragged calls: the group stride is the sum of the lengths rounded up to this many rows (MMDM_RAG_BUCKET)
    int *d_rag = nullptr;                                  // ragged row maps: item_off | item_len | row_item | row_pos | row_seq | seq_off | seq_len
    int *d_item_off = nullptr, *d_item_len = nullptr, *d_row_item = nullptr, *d_row_pos = nullptr, *d_row_seq = nullptr, *d_seq_off = nullptr, *d_seq_len = nullptr, *d_item_order = nullptr;

    // workspace
    Scratch sa, sb;                                        // sa: denoiser1 + Influence, sb: denoiser2 (runs concurrently)
    float *mI = nullptr;                                   // Influence CA source  [R, Dm]
    float *o1 = nullptr, *o2 = nullptr, *out1 = nullptr, *out2 = nullptr;   // [n,T,524]
    float *w23 = nullptr, *hpool = nullptr;
    float *model_out = nullptr, *x = nullptr, *x2 = nullptr, *px1 = nullptr, *px2 = nullptr, *floor_ws = nullptr;
    float *cond_cat = nullptr;                             // [n, 8*text_dim]
    float *txt_d1 = nullptr, *txt_d2 = nullptr, *txt_mx = nullptr;      // text_embed outputs
    float *se_d1 = nullptr, *se_d2 = nullptr, *se_mx = nullptr;         // silu(time + text)
    float *ss_d1 = nullptr, *ss_d2 = nullptr, *ss_mx = nullptr;         // AdaLN (scale|shift) for every layer/norm
    float *tt_tmp = nullptr, *tt_tmp2 = nullptr;           // [Smax, maxD] schedule scratch
    float *tt_stash = nullptr;                             // [3, maxD] row 0 of the three time tables while mmdm_module_forward borrows it
    float *dual_w = nullptr;                               // [Smax] DualMDM composition weight per respaced step (single_only == 3)
    bool dual_w_set = false;
    int td1 = 0, cond_w = 0;                               // denoiser1's cond width (text_dim, or the latent size for MDM) and the cond row width

    // history: host mirror + the device-side descriptor the step's kernels read (kernels.h: mmdm_hist_desc)
    mmdm_hist_desc hist = {nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0};
    mmdm_hist_desc* d_hist = nullptr;
    hipStream_t call_stream = nullptr;      // stream of the last mmdm_begin: mmdm_set_history orders its descriptor update on it

    // Captured step graphs, least-recently-used cache keyed by everything a captured node bakes in: (B, T, S).  History
    // destinations, schedule tables, conditioning and the step index are device-side data, not node arguments, so the eval
    // caller's alternating (B, T) requests (src/evaluation/datasets.py:101-122, 438) replay cached graphs instead of re-capturing.
    // (ragged calls: T = query tiles of the longest item, rows = the group stride; uniform calls: rows = 0)
    struct GraphEntry { int B, T, S, rows; hipGraphExec_t exec; uint64_t used; hipEvent_t done; };      // done: recorded behind the entry's last replay
    std::vector<GraphEntry> graphs;
    size_t graph_cap = 8;
    uint64_t graph_clock = 0;
    int64_t n_captures = 0, n_replays = 0;
    int device = 0;

    // denoiser1 || denoiser2 on two streams: the two stacks are independent until the mixer (mixermdm.py:685-687), so the
    // tail of one model's kernels overlaps the other's and the HBM-bound kernels hide under MFMA-bound ones.
    hipStream_t st2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr;
    bool overlap = true;
    // switches read from the environment ONCE, at mmdm_create (include/mmdm.h lists them): a handle's behaviour never changes afterwards
    bool force_qkp = false, no_qkp = false, no_pvb = false, no_pack = false;      // MMDM_QKP, MMDM_NO_QKP, MMDM_NO_BF16_PV, MMDM_NO_PACK
    bool no_split_embed = false;                                                  // MMDM_NO_SPLIT_EMBED
    bool no_split_cond = false;                                                   // MMDM_NO_SPLIT_COND

    Prof prof;
};

namespace {

int herr(mmdm_handle h, int code) {
    if (code) snprintf(h->err, sizeof(h->err), "%s", g_err);
    return code;
}

int dalloc(mmdm_handle h, float** p, size_t nfloats) {
    void* q = nullptr;
    if (nfloats == 0) nfloats = 1;
    HIPCHK(hipMalloc(&q, nfloats * sizeof(float)));
    (h->alloc_weights ? h->wb->allocs : h->allocs).push_back(q);
    *p = static_cast<float*>(q);
    return MMDM_OK;
}

int add_slot(mmdm_handle h, const std::string& name, float** p, int64_t rows, int64_t cols, int64_t ld = 0, bool zero = false) {
    if (ld == 0) ld = cols;
    RC(dalloc(h, p, (size_t)rows * ld));
    if (zero || ld != cols) HIPCHK(hipMemset(*p, 0, (size_t)rows * ld * sizeof(float)));
    Slot s;
    s.dst = *p; s.rows = rows; s.cols = cols; s.ld = ld;
    h->slots[name] = s;
    return MMDM_OK;
}

// slot that points into an already allocated packed buffer
void add_view(mmdm_handle h, const std::string& name, float* p, int64_t rows, int64_t cols) {
    Slot s;
    s.dst = p; s.rows = rows; s.cols = cols; s.ld = cols;
    h->slots[name] = s;
}

void add_ignored(mmdm_handle h, const std::string& name) {
    Slot s;
    s.required = false;
    h->slots[name] = s;
}

int build_stack(mmdm_handle h, StackW& st, const std::string& pfx, int D, int F, int L, int H, bool has_ca, bool ca_keys_ignored) {
    st.D = D; st.F = F; st.L = L; st.H = H; st.has_ca = has_ca; st.n_ada = has_ca ? 4 : 2;
    RC(dalloc(h, &st.ada_w, (size_t)L * st.n_ada * 2 * D * D));
    RC(dalloc(h, &st.ada_b, (size_t)L * st.n_ada * 2 * D));
    if (h->cfg.precision >= 1 && D % 64 == 0) {       // two fp16 planes [2][L*n_ada*2D][D]: as many bytes as the fp32 matrix
        float* q = nullptr;
        RC(dalloc(h, &q, (size_t)L * st.n_ada * 2 * D * D));
        st.ada_s = q;
    }
    st.layers.resize(L);
    const bool bf = h->cfg.precision >= 1;
    const size_t planes = h->cfg.precision == 2 ? MMDM_SPLIT_NPL : 1;   // fp32-split: two fp16 planes per weight (gemm_split.hip)
    if (bf) st.layers_b.resize(L);
    auto twin = [&](void** p, size_t n) -> int {          // n 16-bit elements per plane
        float* q = nullptr;
        RC(dalloc(h, &q, (planes * n + 1) / 2));
        *p = q;
        return MMDM_OK;
    };
    auto twin8 = [&](void** p, float** sc, size_t rows, size_t cols) -> int {      // fp8 bytes [rows, cols] + fp32 scale per row
        float* q = nullptr;
        RC(dalloc(h, &q, (rows * cols + 3) / 4));
        *p = q;
        return dalloc(h, sc, rows);
    };
    for (int i = 0; i < L; ++i) {
        LayerW& lw = st.layers[i];
        memset(&lw, 0, sizeof(lw));
        if (h->cfg.precision == 3) {
            LayerWB& b2 = st.layers_b[i];
            RC(twin8(&b2.sa_in_8, &b2.sa_in_s, (size_t)3 * D, D));
            RC(twin(&b2.sa_out_w, (size_t)D * D));
            RC(twin8(&b2.f1_8, &b2.f1_s, F, D));
            RC(twin8(&b2.f2_8, &b2.f2_s, D, F));
            if (has_ca) {
                RC(twin8(&b2.ca_in_8, &b2.ca_in_s, (size_t)3 * D, D));
                RC(twin(&b2.ca_out_w, (size_t)D * D));
            }
        } else if (bf) {
            LayerWB& b2 = st.layers_b[i];
            RC(twin(&b2.sa_in_w, (size_t)3 * D * D));
            RC(twin(&b2.sa_out_w, (size_t)D * D));
            RC(twin(&b2.f1_w, (size_t)F * D));
            RC(twin(&b2.f2_w, (size_t)D * F));
            if (has_ca) {
                RC(twin(&b2.ca_in_w, (size_t)3 * D * D));
                RC(twin(&b2.ca_out_w, (size_t)D * D));
            }
        }
        const std::string b = pfx + "blocks." + std::to_string(i) + ".";
        auto ada = [&](const char* norm, int slot) {
            add_view(h, b + norm + ".emb_layers.1.weight", st.ada_w + ((size_t)i * st.n_ada + slot) * 2 * D * D, 2 * D, D);
            add_view(h, b + norm + ".emb_layers.1.bias", st.ada_b + ((size_t)i * st.n_ada + slot) * 2 * D, 2 * D, 1);
        };
        ada("sa_block.norm", 0);
        ada("ffn.norm", has_ca ? 3 : 1);
        RC(add_slot(h, b + "sa_block.attention.in_proj_weight", &lw.sa_in_w, 3 * D, D));
        RC(add_slot(h, b + "sa_block.attention.in_proj_bias", &lw.sa_in_b, 3 * D, 1));
        RC(add_slot(h, b + "sa_block.attention.out_proj.weight", &lw.sa_out_w, D, D));
        RC(add_slot(h, b + "sa_block.attention.out_proj.bias", &lw.sa_out_b, D, 1));
        RC(add_slot(h, b + "ffn.linear1.weight", &lw.f1_w, F, D));
        RC(add_slot(h, b + "ffn.linear1.bias", &lw.f1_b, F, 1));
        RC(add_slot(h, b + "ffn.linear2.weight", &lw.f2_w, D, F));
        RC(add_slot(h, b + "ffn.linear2.bias", &lw.f2_b, D, 1));
        if (has_ca) {
            ada("ca_block.norm", 1);
            ada("ca_block.xf_norm", 2);
            RC(add_slot(h, b + "ca_block.attention.in_proj_weight", &lw.ca_in_w, 3 * D, D));
            RC(add_slot(h, b + "ca_block.attention.in_proj_bias", &lw.ca_in_b, 3 * D, 1));
            RC(add_slot(h, b + "ca_block.attention.out_proj.weight", &lw.ca_out_w, D, D));
            RC(add_slot(h, b + "ca_block.attention.out_proj.bias", &lw.ca_out_b, D, 1));
        } else if (ca_keys_ignored) {
            // individual mode carries unused cross-attention weights in its state_dict (blocks.py:45-47, 55-56)
            for (const char* k : {"ca_block.norm.emb_layers.1.weight", "ca_block.norm.emb_layers.1.bias", "ca_block.xf_norm.emb_layers.1.weight",
                                  "ca_block.xf_norm.emb_layers.1.bias", "ca_block.attention.in_proj_weight", "ca_block.attention.in_proj_bias",
                                  "ca_block.attention.out_proj.weight", "ca_block.attention.out_proj.bias"})
                add_ignored(h, b + k);
        }
    }
    return MMDM_OK;
}

int build_module(mmdm_handle h, ModuleW& m, const std::string& pfx, const std::string& stack_pfx, int D, int F, int L, int H,
                 bool has_ca, bool ca_ignored, const std::string& out_name, int out_rows) {
    const int td = h->cfg.text_dim;
    RC(build_stack(h, m.st, stack_pfx, D, F, L, H, has_ca, ca_ignored));
    RC(add_slot(h, pfx + "sequence_pos_encoder.pe", &m.pe, 5000, D));
    add_ignored(h, pfx + "embed_timestep.sequence_pos_encoder.pe");
    RC(add_slot(h, pfx + "motion_embed.weight", &m.me_w, D, NF, NFP));
    RC(add_slot(h, pfx + "motion_embed.bias", &m.me_b, D, 1));
    if (h->cfg.precision >= 1 && D % 128 == 0) {      // the embedding of the low-precision modes runs on the fp32-split kernel (embed())
        float* q = nullptr;
        RC(dalloc(h, &q, (size_t)D * NFS));           // two fp16 planes [2][D][NFS]
        m.me_s = q;
    }
    RC(add_slot(h, pfx + "text_embed.weight", &m.te_w, D, td));
    RC(add_slot(h, pfx + "text_embed.bias", &m.te_b, D, 1));
    RC(add_slot(h, pfx + "embed_timestep.time_embed.0.weight", &m.t0_w, D, D));
    RC(add_slot(h, pfx + "embed_timestep.time_embed.0.bias", &m.t0_b, D, 1));
    RC(add_slot(h, pfx + "embed_timestep.time_embed.2.weight", &m.t2_w, D, D));
    RC(add_slot(h, pfx + "embed_timestep.time_embed.2.bias", &m.t2_b, D, 1));
    RC(add_slot(h, out_name + ".weight", &m.out_w, out_rows, D));
    RC(add_slot(h, out_name + ".bias", &m.out_b, out_rows, 1));
    return MMDM_OK;
}

// MDMDenoiser (src/models/mdm.py:234-298): pose embedding, cond token, post-norm nn.TransformerEncoder, pose head.
int build_module_mdm(mmdm_handle h, ModuleW& m, const std::string& pfx, int D, int F, int L, int H) {
    m.kind = 1;
    m.st.D = D; m.st.F = F; m.st.L = L; m.st.H = H; m.st.n_ada = 0; m.st.has_ca = false;
    RC(add_slot(h, pfx + "sequence_pos_encoder.pe", &m.pe, 5000, D));
    add_ignored(h, pfx + "embed_timestep.sequence_pos_encoder.pe");
    RC(add_slot(h, pfx + "input_process.poseEmbedding.weight", &m.me_w, D, NF, NFP));
    RC(add_slot(h, pfx + "input_process.poseEmbedding.bias", &m.me_b, D, 1));
    RC(add_slot(h, pfx + "embed_timestep.time_embed.0.weight", &m.t0_w, D, D));
    RC(add_slot(h, pfx + "embed_timestep.time_embed.0.bias", &m.t0_b, D, 1));
    RC(add_slot(h, pfx + "embed_timestep.time_embed.2.weight", &m.t2_w, D, D));
    RC(add_slot(h, pfx + "embed_timestep.time_embed.2.bias", &m.t2_b, D, 1));
    RC(add_slot(h, pfx + "output_process.poseFinal.weight", &m.out_w, NF, D));
    RC(add_slot(h, pfx + "output_process.poseFinal.bias", &m.out_b, NF, 1));
    m.enc.resize(L);
    for (int i = 0; i < L; ++i) {
        EncLayerW& e = m.enc[i];
        const std::string b = pfx + "seqTransEncoder.layers." + std::to_string(i) + ".";
        RC(add_slot(h, b + "self_attn.in_proj_weight", &e.in_w, 3 * D, D));
        RC(add_slot(h, b + "self_attn.in_proj_bias", &e.in_b, 3 * D, 1));
        RC(add_slot(h, b + "self_attn.out_proj.weight", &e.out_w, D, D));
        RC(add_slot(h, b + "self_attn.out_proj.bias", &e.out_b, D, 1));
        RC(add_slot(h, b + "linear1.weight", &e.l1_w, F, D));
        RC(add_slot(h, b + "linear1.bias", &e.l1_b, F, 1));
        RC(add_slot(h, b + "linear2.weight", &e.l2_w, D, F));
        RC(add_slot(h, b + "linear2.bias", &e.l2_b, D, 1));
        RC(add_slot(h, b + "norm1.weight", &e.n1_g, D, 1));
        RC(add_slot(h, b + "norm1.bias", &e.n1_b, D, 1));
        RC(add_slot(h, b + "norm2.weight", &e.n2_g, D, 1));
        RC(add_slot(h, b + "norm2.bias", &e.n2_b, D, 1));
    }
    return MMDM_OK;
}

// ------------------------------------------------------------------------------------------------------
// profiled launches
// ------------------------------------------------------------------------------------------------------
struct Ctx {
    mmdm_handle h;
    hipStream_t st;
    const Scratch* s;
    const Geom* g = nullptr;      // row geometry (nullptr: uniform, sizes as passed)
    bool rag() const { return g && g->rag; }
    size_t rows_of(int nseq, int T) const { return g ? g->rows_of(nseq) : (size_t)nseq * T; }
};

// the row-split rule of the fp32 GEMM dispatch is thread-local state: set for the duration of one step / module forward, then restored
struct TailScope {
    int prev;
    explicit TailScope(int t) : prev(mmdm_gemm_get_tail()) { mmdm_gemm_set_tail(t); }
    ~TailScope() { mmdm_gemm_set_tail(prev); }
};

int prof_begin(const Ctx& c, int cls, double flops, double bytes = 0) {
    if (!c.h) return MMDM_OK;                 // stateless C-ABI composites run without a handle
    Prof& p = c.h->prof;
    if (!p.on) return MMDM_OK;
    if (p.used[cls] + 2 > p.ev[cls].size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            p.ev[cls].push_back(e);
        }
    }
    p.flops[cls] += flops;
    p.bytes[cls] += bytes;
    HIPCHK(hipEventRecord(p.ev[cls][p.used[cls]], c.st));
    return MMDM_OK;
}

int prof_end(const Ctx& c, int cls) {
    if (!c.h) return MMDM_OK;
    Prof& p = c.h->prof;
    if (!p.on) return MMDM_OK;
    HIPCHK(hipEventRecord(p.ev[cls][p.used[cls] + 1], c.st));
    p.used[cls] += 2;
    return MMDM_OK;
}

int linear(const Ctx& c, const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K,
           int epi = MMDM_EPI_BIAS, const float* extra = nullptr, int ld_extra = 0, int period = 0, int Kw = 0) {
    const int cls = c.h && c.h->cfg.precision != 0 ? 3 : 0;
    RC(prof_begin(c, cls, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N * (epi == MMDM_EPI_BIAS_RESID ? 2 : 1))));
    RC(mmdm_linear_f32_ex(A, lda, W, ldw, Kw ? Kw : K, bias, C, ldc, M, N, K, epi, extra, ld_extra, period, c.st));
    return prof_end(c, cls);
}

// plain self-attention of nn.TransformerEncoderLayer (no zero key)
int attention_plain(const Ctx& c, const float* qkv, int ld, float* O, int ldo, int nseq, int T, int H, int dh, int flags) {
    RC(prof_begin(c, 1, 4.0 * nseq * H * (double)T * T * dh, 4.0 * nseq * H * dh * 4.0 * T));
    RC(mmdm_attention_opts(qkv, ld, qkv + H * dh, ld, qkv + 2 * H * dh, ld, O, ldo, 0, flags, nseq, T, T, H, dh, 0, c.st));
    return prof_end(c, 1);
}

// ------------------------------------------------------------------------------------------------------
// transformer stack
// ------------------------------------------------------------------------------------------------------
struct StackRun {
    int nseq, T;
    const float* ss;        // [rows, ss_ld] AdaLN projections of every layer/norm
    int ss_ld;
    int sa_row0, sa_rows;   // rows of `ss` used by sa_block.norm (row = row0 + s % rows)
    int ca_row0, ca_rows;   // ... by ca_block.norm and ca_block.xf_norm
    int ffn_row0, ffn_rows; // ... by ffn.norm
    int ca_mode;            // 0 none; 1 keys/values = the other half of the layer INPUT (in2in.py:439-440); 2 = fixed `kv_src`
    const float* kv_src;
    int l0 = 0;             // first block to run (the "dual_individual" quirk runs only the last block on person b)
};

// h [nseq*T, D] is updated in place through the L blocks (TransformerBlockDoubleCond / TransformerBlock / InfluenceBlockCross).
// bf16-operand GEMM with the same accounting as linear()
struct Second {           // optional second GEMM output: leading columns also as bf16 plane(s) for the attention kernel
    void* p = nullptr;
    int ld = 0, cols = 0;
    size_t plane = 0;
};

int linear_b(const Ctx& c, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int out_bf16, int M, int N, int K,
             int epi = MMDM_EPI_BIAS, const float* extra = nullptr, int ld_extra = 0, Second s2 = Second()) {
    RC(prof_begin(c, 0, 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K) + (out_bf16 ? 2.0 : 4.0) * M * N * (epi == MMDM_EPI_BIAS_RESID ? 2 : 1)));
    RC(mmdm_linear_bf16_ex(A, lda, W, ldw, bias, C, ldc, out_bf16, M, N, K, epi, extra, ld_extra, 0, s2.p, s2.ld, s2.cols, c.st));
    return prof_end(c, 0);
}

// fp8-operand GEMM (precision == 3): A fp8 + per-row scales (nullptr = unit), W fp8 + per-output-channel scales (gemm_bf16.hip, ET = 1)
int linear_8(const Ctx& c, const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, const float* bias, void* C, int ldc,
             int out_mode, int M, int N, int K, int epi, const float* extra, int ld_extra, Second s2 = Second(), float a_const = 1.f, float out_scale = 1.f) {
    RC(prof_begin(c, 2, 2.0 * M * N * K, 1.0 * ((double)M * K + (double)N * K) + (out_mode == 0 ? 4.0 : out_mode == 1 ? 2.0 : 1.0) * M * N * (epi == MMDM_EPI_BIAS_RESID ? 2 : 1)));
    RC(mmdm_linear_fp8_ex(A, lda, a_scale, W, ldw, w_scale, bias, C, ldc, out_mode, M, N, K, epi, extra, ld_extra, 0, s2.p, s2.ld, s2.cols, a_const, out_scale, c.st));
    return prof_end(c, 2);
}

// fp32-split GEMM (precision == 2): A and W as two fp16 planes, fp32 accuracy on the 16-bit matrix cores (gemm_split.hip)
int linear_s(const Ctx& c, const void* A, int lda, size_t a_plane, const void* W, int ldw, size_t w_plane, const float* bias, void* C, int ldc,
             size_t c_plane, int out_split, int M, int N, int K, int epi, const float* extra, int ld_extra, Second s2 = Second(), int period = 0) {
    const int cls = c.h && c.h->cfg.precision != 2 ? 3 : 0;       // in a bf16 / fp8 handle this is one of the fp32-accurate side GEMMs
    RC(prof_begin(c, cls, 2.0 * M * N * K, 2.0 * MMDM_SPLIT_NPL * ((double)M * K + (double)N * K) + 4.0 * M * N * (epi == MMDM_EPI_BIAS_RESID ? 2 : 1)));
    RC(mmdm_linear_split_ex(A, lda, (int64_t)a_plane, W, ldw, (int64_t)w_plane, bias, C, ldc, (int64_t)c_plane, out_split, M, N, K, epi, extra, ld_extra, period,
                            s2.p, s2.ld, (int64_t)s2.plane, s2.cols, c.st));
    return prof_end(c, cls);
}

// attention whose Q K^T runs on the bf16 matrix cores from the plane copies written by the projection GEMMs
// algorithmic work of one attention launch: 4 dh (T (T + 1)) per (sequence, head) -- summed over the items' own lengths in a ragged batch
double attn_flops(const Ctx& c, int nseq, int H, int Tq, int Tk, int dh) {
    if (c.rag()) return 4.0 * (nseq / c.g->B) * H * dh * c.g->tt1;
    return 4.0 * nseq * H * (double)Tq * (Tk + 1) * dh;
}
double attn_bytes(const Ctx& c, int nseq, int H, int Tq, int Tk, int dh) {
    if (c.rag()) return 4.0 * (nseq / c.g->B) * H * dh * 4.0 * c.g->real_rows;
    return 4.0 * nseq * H * dh * (2.0 * Tq + 2.0 * Tk);
}
// the ragged call's sequence description for the attention kernels (nullptr in the uniform layout); `nseq` sequences = nseq / B groups
const mmdm_rag_seq* rag_seq(const Ctx& c, int nseq, mmdm_rag_seq& tmp) {
    if (!c.rag()) return nullptr;
    tmp = mmdm_rag_seq{c.g->seq_off, c.g->seq_len, (int)c.g->rows_of(nseq), c.g->T, c.g->item_order, c.g->B};
    return &tmp;
}

int attention_p(const Ctx& c, const void* Qp, int ldq, size_t q_plane, const void* Kp, int ldk, size_t k_plane, int np, const float* V, int ldv, void* O, int ldo,
                int out_mode, int nseq, int Tq, int Tk, int H, int dh, int shift, const void* Vp = nullptr, int ldvp = 0, size_t v_plane = 0) {
    mmdm_rag_seq tmp;
    RC(prof_begin(c, 1, attn_flops(c, nseq, H, Tq, Tk, dh), attn_bytes(c, nseq, H, Tq, Tk, dh)));
    RC(mmdm_attention_planes_ex(Qp, ldq, (int64_t)q_plane, Kp, ldk, (int64_t)k_plane, np, V, ldv, Vp, ldvp, (int64_t)v_plane, O, ldo, out_mode, 0, nseq, Tq, Tk, H, dh, shift, c.st,
                                rag_seq(c, nseq, tmp)));
    return prof_end(c, 1);
}

int attention_b(const Ctx& c, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* O, int ldo, int out_bf16,
                int nseq, int Tq, int Tk, int H, int dh, int shift) {
    mmdm_rag_seq tmp;
    RC(prof_begin(c, 1, attn_flops(c, nseq, H, Tq, Tk, dh), attn_bytes(c, nseq, H, Tq, Tk, dh)));
    RC(mmdm_attention_opts_rag(Q, ldq, K, ldk, V, ldv, O, ldo, out_bf16, 0, nseq, Tq, Tk, H, dh, shift, rag_seq(c, nseq, tmp), c.st));
    return prof_end(c, 1);
}

int ss_ld_of(const ModuleW& m) { return m.st.L * m.st.n_ada * 2 * m.st.D; }

// h [nseq*T, D] is updated in place through the L blocks (TransformerBlockDoubleCond / TransformerBlock / InfluenceBlockCross).
// precision == 1: the GEMM operands xn / att / f1 are written as bf16 by their producers and the weights come from the bf16 twins;
// the residual stream h, the Q/K/V projections, softmax and all accumulation stay fp32.
int run_stack(const Ctx& c, const StackW& w, float* hbuf, const StackRun& r) {
    const Scratch& S = *c.s;
    const int D = w.D, F = w.F, R = (int)c.rows_of(r.nseq, r.T), dh = D / w.H;
    const int prec = c.h->cfg.precision;
    const int* const row_seq = c.rag() ? c.g->row_seq : nullptr;      // ragged: the AdaLN kernel looks a row's sequence up instead of dividing by T
    const bool bf = prec >= 1, f8 = prec == 3;
    const int ob = f8 ? 1 : prec;           // output mode of the attention (operand of the out-projection): 0 fp32, 1 bf16, 2 the two fp16 split planes
    auto bw = [](const void* base, size_t elems) { return static_cast<const void*>(static_cast<const uint16_t*>(base) + elems); };
    // one GEMM of the stack: fp32 (A fp32, W fp32), bf16 (A bf16 from the producer, W twin) or fp32-split (two fp16 planes each).
    // wtot = elements of the whole weight matrix the twin was made from (its plane stride); the A plane stride is R*K.
    auto gemm = [&](const float* A, int lda, const float* Wf, const void* Wb, size_t woff, size_t wtot, const float* bias, float* C, int ldc, int out_b,
                    int N, int K, int epi, const float* extra, int ld_extra, Second s2 = Second()) -> int {
        if (prec == 2) return linear_s(c, A, lda, (size_t)R * K, bw(Wb, woff), w.w_packed ? 0 : K, wtot, bias, C, ldc, (size_t)R * N, out_b == 2, R, N, K, epi, extra, ld_extra, s2);
        if (bf) return linear_b(c, A, lda, bw(Wb, woff), w.w_packed ? 0 : K, bias, C, ldc, out_b, R, N, K, epi, extra, ld_extra, s2);
        return linear(c, A, lda, Wf + woff, K, bias, C, ldc, R, N, K, epi, extra, ld_extra);
    };
    // bf16 path with a head size the plane kernel covers: the projection GEMMs also emit a bf16 copy of Q and K and the scores come from
    // the bf16 matrix cores (attn_qkp_kernel): 20.1 -> 18.7 ms/step.
    // fp32-split mode: the projections are written as the two fp16 planes INSTEAD of fp32 rows (same bytes) and the attention works on the planes
    // throughout (attn_qkp_kernel<DH, 2, true, true>); MMDM_NO_QKP=1 keeps fp32 projections + the fp32 attention kernel, MMDM_QKP=1 selects the
    // round-2 experiment (extra three-way bf16 copies of Q | K for the scores only: measured slower than the fp32 kernel, 51.6 vs 49.8 ms/step).
    const bool spa = prec == 2 && (dh == 64 || dh == 128) && !c.h->no_qkp && !c.h->force_qkp;
    const bool qkp = bf && (prec == 1 || prec == 3 || c.h->force_qkp) && (dh == 64 || dh == 128) && S.qk && !c.h->no_qkp;
    const int np = prec == 2 ? 3 : 1;
    // precision 3: one fp8 GEMM of the stack.  W8 / Ws: fp8 matrix [rows, K] and its per-output-channel scales, row0 = first output
    // channel used (the K|V slice of the packed cross-attention projection); unit_a: A is the GELU output, stored at the static scale GSCALE
    constexpr float GSCALE = 1.0f;       // (a static x16 on the GELU tensor was measured: its range up to 28 saturates in these networks and doubles the error)
    auto gemm8 = [&](const void* A, bool unit_a, const void* W8, const float* Ws, size_t row0, const float* bias, void* C, int ldc, int out_mode, int N, int K,
                     int epi, const float* extra, int ld_extra, Second s2 = Second(), bool w_frag = false) -> int {
        return linear_8(c, A, K, unit_a ? nullptr : S.xs, static_cast<const uint8_t*>(W8) + row0 * K, w_frag ? 0 : K, Ws + row0, bias, C, ldc, out_mode, R, N, K, epi, extra, ld_extra, s2,
                        unit_a ? 1.0f / GSCALE : 1.0f, out_mode == 2 ? GSCALE : 1.0f);
    };
    // AdaLN into the stack GEMMs' operand format: fp32 / bf16 / two fp16 planes, or fp8 + per-row scales
    auto norm = [&](const float* src, const float* ssp, int rows) -> int {
        return ROWOP(c, mmdm_adaln_any, src, ssp, r.ss_ld, rows, S.xn, f8 ? 3 : ob, f8 ? S.xs : nullptr, r.nseq, r.T, D, c.st, row_seq, R);
    };
    auto second = [&](void* buf, int ld, int cols) { Second s2; if (qkp) { s2.p = buf; s2.ld = ld; s2.cols = cols; s2.plane = (size_t)R * ld; } return s2; };
    // one-plane (bf16 / fp8) modes: the projection GEMMs' bf16 copy also covers V, and P.V runs on the bf16 matrix cores (attn_qkp_kernel<DH, 1, true>)
    const bool pvb = qkp && np == 1 && !c.h->no_pvb;
    for (int l = r.l0; l < w.L; ++l) {
        const LayerW& lw = w.layers[l];
        const LayerWB lb = bf ? w.layers_b[l] : LayerWB();
        auto ss_at = [&](int slot, int row0) { return r.ss + (size_t)row0 * r.ss_ld + ((size_t)l * w.n_ada + slot) * 2 * D; };
        // --- self attention (layers.py:36-45)
        RC(norm(hbuf, ss_at(0, r.sa_row0), r.sa_rows));
        const int qkld = pvb ? 3 * D : 2 * D;           // row stride of the bf16 copy of the packed projection: Q|K or Q|K|V
        // all-bf16 attention: nothing reads the fp32 projection, so the GEMM writes bf16 only (a third of the bytes: the fp8 QKV GEMM is output-bound)
        const _Float16* const qkvh = reinterpret_cast<const _Float16*>(S.qkv);      // spa: planes [2][R][3D] (self) / [2][R][D] (cross-attention queries)
        const _Float16* const kvh = reinterpret_cast<const _Float16*>(S.kv);        // spa: planes [2][R][2D]
        if (spa) RC(gemm(S.xn, D, lw.sa_in_w, lb.sa_in_w, 0, (size_t)3 * D * D, lw.sa_in_b, S.qkv, 3 * D, 2, 3 * D, D, MMDM_EPI_BIAS, nullptr, 0));
        else if (pvb) {
            if (f8) RC(gemm8(S.xn, false, lb.sa_in_8, lb.sa_in_s, 0, lw.sa_in_b, S.qk, 3 * D, 1, 3 * D, D, MMDM_EPI_BIAS, nullptr, 0, Second(), w.w_packed));
            else RC(gemm(S.xn, D, lw.sa_in_w, lb.sa_in_w, 0, (size_t)3 * D * D, lw.sa_in_b, static_cast<float*>(S.qk), 3 * D, 1, 3 * D, D, MMDM_EPI_BIAS, nullptr, 0));
        } else if (f8) RC(gemm8(S.xn, false, lb.sa_in_8, lb.sa_in_s, 0, lw.sa_in_b, S.qkv, 3 * D, 0, 3 * D, D, MMDM_EPI_BIAS, nullptr, 0, second(S.qk, qkld, qkld), w.w_packed));
        else RC(gemm(S.xn, D, lw.sa_in_w, lb.sa_in_w, 0, (size_t)3 * D * D, lw.sa_in_b, S.qkv, 3 * D, 0, 3 * D, D, MMDM_EPI_BIAS, nullptr, 0, second(S.qk, qkld, qkld)));
        if (spa) RC(attention_p(c, qkvh, 3 * D, (size_t)R * 3 * D, qkvh + D, 3 * D, (size_t)R * 3 * D, 2, nullptr, 0, S.att, D, ob, r.nseq, r.T, r.T, w.H, dh, 0,
                                qkvh + 2 * D, 3 * D, (size_t)R * 3 * D));
        else if (qkp) RC(attention_p(c, S.qk, qkld, (size_t)R * qkld, static_cast<const uint16_t*>(S.qk) + D, qkld, (size_t)R * qkld, np, S.qkv + 2 * D, 3 * D, S.att, D, ob,
                                r.nseq, r.T, r.T, w.H, dh, 0, pvb ? static_cast<const uint16_t*>(S.qk) + 2 * D : nullptr, qkld));
        else RC(attention_b(c, S.qkv, 3 * D, S.qkv + D, 3 * D, S.qkv + 2 * D, 3 * D, S.att, D, ob, r.nseq, r.T, r.T, w.H, dh, 0));
        if (r.ca_mode) {
            // keys/values of the cross attention come from the layer INPUT of the other stream (or a fixed source):
            // project them before the residual below overwrites h.
            const float* src = r.ca_mode == 1 ? hbuf : r.kv_src;
            RC(norm(src, ss_at(2, r.ca_row0), r.ca_rows));
            const int kvld = pvb ? 2 * D : D;             // bf16 copy of the cross-attention projection: K or K|V
            if (spa) RC(gemm(S.xn, D, lw.ca_in_w, lb.ca_in_w, (size_t)D * D, (size_t)3 * D * D, lw.ca_in_b + D, S.kv, 2 * D, 2, 2 * D, D, MMDM_EPI_BIAS, nullptr, 0));
            else if (pvb) {
                if (f8) RC(gemm8(S.xn, false, lb.ca_in_8, lb.ca_in_s, D, lw.ca_in_b + D, S.kvp, 2 * D, 1, 2 * D, D, MMDM_EPI_BIAS, nullptr, 0, Second(), w.w_packed));
                else RC(gemm(S.xn, D, lw.ca_in_w, lb.ca_in_w, (size_t)D * D, (size_t)3 * D * D, lw.ca_in_b + D, static_cast<float*>(S.kvp), 2 * D, 1, 2 * D, D, MMDM_EPI_BIAS, nullptr, 0));
            } else if (f8) RC(gemm8(S.xn, false, lb.ca_in_8, lb.ca_in_s, D, lw.ca_in_b + D, S.kv, 2 * D, 0, 2 * D, D, MMDM_EPI_BIAS, nullptr, 0, second(S.kvp, kvld, kvld), w.w_packed));
            else RC(gemm(S.xn, D, lw.ca_in_w, lb.ca_in_w, (size_t)D * D, (size_t)3 * D * D, lw.ca_in_b + D, S.kv, 2 * D, 0, 2 * D, D, MMDM_EPI_BIAS, nullptr, 0, second(S.kvp, kvld, kvld)));
        }
        const int ffn_slot = w.has_ca ? 3 : 1;
        if (r.ca_mode) RC(gemm(S.att, D, lw.sa_out_w, lb.sa_out_w, 0, (size_t)D * D, lw.sa_out_b, hbuf, D, 0, D, D, MMDM_EPI_BIAS_RESID, hbuf, D));
        else RC(gemm(S.att, D, lw.sa_out_w, lb.sa_out_w, 0, (size_t)D * D, lw.sa_out_b, hbuf, D, 0, D, D, MMDM_EPI_BIAS_RESID, hbuf, D));
        // --- cross attention (layers.py:77-88)
        if (r.ca_mode) {
            RC(norm(hbuf, ss_at(1, r.ca_row0), r.ca_rows));
            if (spa) RC(gemm(S.xn, D, lw.ca_in_w, lb.ca_in_w, 0, (size_t)3 * D * D, lw.ca_in_b, S.qkv, D, 2, D, D, MMDM_EPI_BIAS, nullptr, 0));
            else if (pvb) {
                if (f8) RC(gemm8(S.xn, false, lb.ca_in_8, lb.ca_in_s, 0, lw.ca_in_b, S.qk, D, 1, D, D, MMDM_EPI_BIAS, nullptr, 0, Second(), w.w_packed));
                else RC(gemm(S.xn, D, lw.ca_in_w, lb.ca_in_w, 0, (size_t)3 * D * D, lw.ca_in_b, static_cast<float*>(S.qk), D, 1, D, D, MMDM_EPI_BIAS, nullptr, 0));
            } else if (f8) RC(gemm8(S.xn, false, lb.ca_in_8, lb.ca_in_s, 0, lw.ca_in_b, S.qkv, D, 0, D, D, MMDM_EPI_BIAS, nullptr, 0, second(S.qk, D, D), w.w_packed));
            else RC(gemm(S.xn, D, lw.ca_in_w, lb.ca_in_w, 0, (size_t)3 * D * D, lw.ca_in_b, S.qkv, D, 0, D, D, MMDM_EPI_BIAS, nullptr, 0, second(S.qk, D, D)));
            if (spa) RC(attention_p(c, qkvh, D, (size_t)R * D, kvh, 2 * D, (size_t)R * 2 * D, 2, nullptr, 0, S.att, D, ob, r.nseq, r.T, r.T, w.H, dh,
                                    r.ca_mode == 1 ? r.nseq / 2 : 0, kvh + D, 2 * D, (size_t)R * 2 * D));
            else if (qkp) RC(attention_p(c, S.qk, D, (size_t)R * D, S.kvp, pvb ? 2 * D : D, (size_t)R * (pvb ? 2 * D : D), np, S.kv + D, 2 * D, S.att, D, ob, r.nseq, r.T, r.T, w.H, dh,
                                    r.ca_mode == 1 ? r.nseq / 2 : 0, pvb ? static_cast<const uint16_t*>(S.kvp) + D : nullptr, 2 * D));
            else RC(attention_b(c, S.qkv, D, S.kv, 2 * D, S.kv + D, 2 * D, S.att, D, ob, r.nseq, r.T, r.T, w.H, dh, r.ca_mode == 1 ? r.nseq / 2 : 0));
            RC(gemm(S.att, D, lw.ca_out_w, lb.ca_out_w, 0, (size_t)D * D, lw.ca_out_b, hbuf, D, 0, D, D, MMDM_EPI_BIAS_RESID, hbuf, D));
        }
        // --- FFN (layers.py:99-106)
        RC(norm(hbuf, ss_at(ffn_slot, r.ffn_row0), r.ffn_rows));
        if (f8) {               // FFN on fp8 operands: the GELU output is written as e4m3 at unit scale and read back as the down-projection's A
            RC(gemm8(S.xn, false, lb.f1_8, lb.f1_s, 0, lw.f1_b, S.f1, F, 2, F, D, MMDM_EPI_BIAS_GELU, nullptr, 0, Second(), w.w_packed));
            RC(gemm8(S.f1, true, lb.f2_8, lb.f2_s, 0, lw.f2_b, hbuf, D, 0, D, F, MMDM_EPI_BIAS_RESID, hbuf, D, Second(), w.w_packed && F >= 2048));
        } else {
            RC(gemm(S.xn, D, lw.f1_w, lb.f1_w, 0, (size_t)F * D, lw.f1_b, S.f1, F, ob, F, D, MMDM_EPI_BIAS_GELU, nullptr, 0));
            RC(gemm(S.f1, F, lw.f2_w, lb.f2_w, 0, (size_t)D * F, lw.f2_b, hbuf, D, 0, D, F, MMDM_EPI_BIAS_RESID, hbuf, D));
        }
    }
    return MMDM_OK;
}

// emb rows -> silu -> all AdaLN projections of a module.  se = silu(time_tab[step] + txt), ss = se W_ada^T + b_ada.
int cond_vectors(const Ctx& c, const ModuleW& m, const float* txt, float* se, float* ss, int rows) {
    const StackW& w = m.st;
    const int N = ss_ld_of(m);
    if (w.ada_s) {       // low-precision handles: `se` holds the two fp16 planes of silu(.) (as many bytes), the projection runs on the fp32-split kernel
        const size_t pa = (size_t)rows * w.D;
        RC(mmdm_cond_silu_planes_nopk(m.time_tab, c.h->d_step, txt, reinterpret_cast<_Float16*>(se), pa, rows, w.D, c.st));
        return linear_s(c, se, w.D, pa, w.ada_s, c.h->no_pack ? w.D : 0, (size_t)N * w.D, w.ada_b, ss, N, 0, 0, rows, N, w.D, MMDM_EPI_BIAS, nullptr, 0);
    }
    RC(ROWOP(c, mmdm_cond_silu_f32, m.time_tab, c.h->d_step, txt, se, rows, w.D, c.st));
    return linear(c, se, w.D, w.ada_w, w.D, w.ada_b, ss, N, rows, N, w.D);
}

// motion_embed + positional encoding of one person slice (in2in.py:426-431): x [nb*T rows, ld 524 or 262] -> h rows
// xp: the repacked pose rows of `repack` below, `rows` = nb*T rows per person; pe_row0 = 1 for MDMDenoiser (token 0 is the conditioning token).
// Low-precision handles (m.me_s): the embedding runs on the fp32-split kernel -- fp32-accurate like the fp32 MFMA kernel it replaces there, at a
// third of its time (K = 262 is 10 steps of the packed kernel); the fp32 mode keeps the fp32 MFMA kernel.
// Ragged call: the PE rows were gathered per frame row of a group at mmdm_begin_ragged (m.pe_r [rows, D]) and the epilogue's `row % period`
// runs over the group stride -- the GEMM kernels are untouched.
int embed(const Ctx& c, const ModuleW& m, const float* xp, int p, float* hdst, int nb, int T, int pe_row0 = 0) {
    const float* pe = c.rag() ? m.pe_r : m.pe + (size_t)pe_row0 * m.st.D;
    const size_t rows = c.rows_of(nb, T);
    const int period = c.rag() ? c.g->rows : T;
    if (m.me_s) {
        const _Float16* a = reinterpret_cast<const _Float16*>(xp) + (size_t)p * 2 * rows * NFS;
        return linear_s(c, a, NFS, rows * NFS, m.me_s, c.h->no_pack ? NFS : 0, (size_t)m.st.D * NFS, m.me_b, hdst, m.st.D, 0, 0, (int)rows, m.st.D, NFS, MMDM_EPI_BIAS_PE, pe,
                        m.st.D, Second(), period);
    }
    return linear(c, xp + (size_t)p * rows * NFP, NFP, m.me_w, NFP, m.me_b, hdst, m.st.D, (int)rows, m.st.D, NFP, MMDM_EPI_BIAS_PE, pe, m.st.D, period, NFP);
}
// pose rows x [rows, ldx] (npers persons side by side) -> the embedding GEMM's A operand: fp32 [npers][rows][NFP], or the two fp16 planes [npers][2][rows][NFS]
int repack(const Ctx& c, const ModuleW& m, const float* x, int ldx, float* xp, int npers, int rows) {
    return m.me_s ? mmdm_repack_pose(x, ldx, xp, npers, rows, NFS, 1, c.st) : mmdm_repack_pose(x, ldx, xp, npers, rows, NFP, 0, c.st);
}

// denoiser1 on the CFG-doubled batch n; xa [B or n rows...]: source rows are taken from `x` with `xrows` samples, repeated to n.
// Sequence order in h: person-major: seq = p*n + b.
int run_denoiser(const Ctx& c, const ModuleW& m, bool interaction, const float* x, int xb, int npers, int ldx, int n, int T,
                 const float* ss, int ss_ld, float* out, int ldo) {
    const int D = m.st.D;
    StackRun r;
    r.nseq = npers * n; r.T = T; r.ss = ss; r.ss_ld = ss_ld;
    r.sa_row0 = 0; r.sa_rows = npers * n;
    r.ffn_row0 = 0; r.ffn_rows = npers * n;
    r.ca_row0 = 2 * n; r.ca_rows = n;
    r.ca_mode = interaction ? 1 : 0;
    r.kv_src = nullptr;
    // embed: `x` holds xb samples (xb == n, or xb == n/2 when cond/uncond halves share the same x: cfg_sampler.py:41-42)
    RC(repack(c, m, x, ldx, c.s->xp, npers, (int)c.rows_of(xb, T)));
    for (int p = 0; p < npers; ++p)
        for (int rep = 0; rep < n / xb; ++rep) {
            const size_t row0 = c.rows_of(p * n + rep * xb, T);
            RC(embed(c, m, c.s->xp, p, c.s->h + row0 * D, xb, T));
        }
    RC(run_stack(c, m.st, c.s->h, r));
    for (int p = 0; p < npers; ++p)   // FinalLayer (layers.py:109-116), per person, concatenated on the channel axis (in2in.py:455-461)
        RC(linear(c, c.s->h + c.rows_of(p * n, T) * D, D, m.out_w, D, m.out_b, out + (size_t)p * NF, ldo, (int)c.rows_of(n, T), NF, D));
    return MMDM_OK;
}

// One nn.TransformerEncoderLayer on x [nseq*T, D] in place.  norm_first = 0 (torch default, MDM / clipTransEncoder):
//   x = LN1(x + SA(x)); x = LN2(x + W2 act(W1 x)).   norm_first = 1 (CLIP ResidualAttentionBlock): x += SA(LN1 x); x += W2 act(W1 LN2 x).
// ws: qkv [R,3D] | att [R,D] | tmp [R,D] | f1 [R,F].
int encoder_layer(const Ctx& c, float* x, const EncLayerW& w, int nseq, int T, int D, int H, int F, bool norm_first, int act_epi, bool causal,
                  float eps, float* qkv, float* att, float* tmp, float* f1) {
    const int R = nseq * T, dh = D / H;
    const int flags = MMDM_ATTN_NO_ZERO_KEY | (causal ? MMDM_ATTN_CAUSAL : 0);
    if (norm_first) {
        RC(ROWOP(c, mmdm_layernorm_f32, x, w.n1_g, w.n1_b, tmp, R, D, eps, c.st));
        RC(linear(c, tmp, D, w.in_w, D, w.in_b, qkv, 3 * D, R, 3 * D, D));
        RC(attention_plain(c, qkv, 3 * D, att, D, nseq, T, H, dh, flags));
        RC(linear(c, att, D, w.out_w, D, w.out_b, x, D, R, D, D, MMDM_EPI_BIAS_RESID, x, D));
        RC(ROWOP(c, mmdm_layernorm_f32, x, w.n2_g, w.n2_b, tmp, R, D, eps, c.st));
        RC(linear(c, tmp, D, w.l1_w, D, w.l1_b, f1, F, R, F, D, act_epi));
        return linear(c, f1, F, w.l2_w, F, w.l2_b, x, D, R, D, F, MMDM_EPI_BIAS_RESID, x, D);
    }
    RC(linear(c, x, D, w.in_w, D, w.in_b, qkv, 3 * D, R, 3 * D, D));
    RC(attention_plain(c, qkv, 3 * D, att, D, nseq, T, H, dh, flags));
    RC(linear(c, att, D, w.out_w, D, w.out_b, tmp, D, R, D, D, MMDM_EPI_BIAS_RESID, x, D));
    RC(ROWOP(c, mmdm_layernorm_f32, tmp, w.n1_g, w.n1_b, x, R, D, eps, c.st));
    RC(linear(c, x, D, w.l1_w, D, w.l1_b, f1, F, R, F, D, act_epi));
    RC(linear(c, f1, F, w.l2_w, F, w.l2_b, tmp, D, R, D, F, MMDM_EPI_BIAS_RESID, x, D));
    return ROWOP(c, mmdm_layernorm_f32, tmp, w.n2_g, w.n2_b, x, R, D, eps, c.st);
}

// MDMDenoiser.forward (mdm.py:273-298) on the CFG-doubled batch: `cond` rows are [n, ldc] with person p's latent-sized slice at
// column p*D; sequence order person-major as in run_denoiser.
int run_denoiser_mdm(const Ctx& c, const ModuleW& m, const float* x, int xb, int npers, int ldx, int n, int T, const float* cond, int ldc,
                     float* out, int ldo) {
    const Scratch& S = *c.s;
    const int D = m.st.D, nseq = npers * n;
    // pose embeddings + pe[1 + t] (token 0 is the conditioning token) into S.att, then assemble [nseq, T+1, D] in S.h
    RC(repack(c, m, x, ldx, S.xp, npers, xb * T));
    for (int p = 0; p < npers; ++p)
        for (int rep = 0; rep < n / xb; ++rep)
            RC(embed(c, m, S.xp, p, S.att + ((size_t)p * n + (size_t)rep * xb) * T * D, xb, T, 1));
    for (int p = 0; p < npers; ++p)
        RC(ROWOP(c, mmdm_mdm_pack, S.att + (size_t)p * n * T * D, cond + (size_t)p * D, ldc, m.time_tab, c.h->d_step, m.pe,
                         S.h + (size_t)p * n * (T + 1) * D, n, T, D, c.st));
    for (int l = 0; l < m.st.L; ++l)
        RC(encoder_layer(c, S.h, m.enc[l], nseq, T + 1, D, m.st.H, m.st.F, false, MMDM_EPI_BIAS_GELU, false, 1e-5f, S.qkv, S.att, S.xn, S.f1));
    RC(ROWOP(c, mmdm_mdm_unpack, S.h, S.att, nseq, T, D, c.st));
    for (int p = 0; p < npers; ++p)
        RC(linear(c, S.att + (size_t)p * n * T * D, D, m.out_w, D, m.out_b, out + (size_t)p * NF, ldo, n * T, NF, D));
    return MMDM_OK;
}

// text_embed of cond slices (in2in.py:415-417, mixermdm.py:677-682): txt[row0 + r] = te(cond[r, col0 : col0+td])
int text_rows(const Ctx& c, const ModuleW& m, const float* cond, int ldc, int col0, float* txt, int row0, int n) {
    const int td = c.h->cfg.text_dim;
    return linear(c, cond + col0, ldc, m.te_w, td, m.te_b, txt + (size_t)row0 * m.st.D, m.st.D, n, m.st.D, td);
}

// in2INDenoiser "dual_individual" (in2in.py:420-422, 441-451): person a runs all blocks; person b's state is never advanced between
// blocks in the reference, so its output is the LAST block applied once to the embedded input.
int run_dual_individual(const Ctx& c, const ModuleW& m, const float* x, int xb, int n, int T, const float* ss, int ss_ld, float* out) {
    const int D = m.st.D;
    RC(repack(c, m, x, NF2, c.s->xp, 2, xb * T));
    for (int p = 0; p < 2; ++p)
        for (int rep = 0; rep < n / xb; ++rep)
            RC(embed(c, m, c.s->xp, p, c.s->h + ((size_t)p * n + (size_t)rep * xb) * T * D, xb, T));
    StackRun r;
    r.nseq = n; r.T = T; r.ss = ss; r.ss_ld = ss_ld;
    r.ca_row0 = 0; r.ca_rows = n; r.ca_mode = 0; r.kv_src = nullptr;
    r.sa_row0 = r.ffn_row0 = 0; r.sa_rows = r.ffn_rows = n; r.l0 = 0;
    RC(run_stack(c, m.st, c.s->h, r));
    r.sa_row0 = r.ffn_row0 = n; r.l0 = m.st.L - 1;
    RC(run_stack(c, m.st, c.s->h + (size_t)n * T * D, r));
    for (int p = 0; p < 2; ++p)
        RC(linear(c, c.s->h + (size_t)p * n * T * D, D, m.out_w, D, m.out_b, out + (size_t)p * NF, NF2, n * T, NF, D));
    return MMDM_OK;
}

int mixer_core(const Ctx& c, int B, int T, bool dyn_hist) {
    // Everything after the two denoisers: mixermdm.py:691-801 + cfg combine.  n = 2B.
    mmdm_handle H = c.h;
    const int n = 2 * B, Dm = H->mx.st.D;
    const mmdm_config& cf = H->cfg;
    const bool rag = c.rag();
    const size_t nT = c.rows_of(n, T);               // frame rows of the CFG-doubled batch (uniform: n * T)
    if (rag) RC(mmdm_mixer_pre_rag(H->o1, H->o2, H->d_stats, H->out1, H->out2, n / B, cf.align, c.g->rg, c.st));
    else RC(mmdm_mixer_pre_f32(H->o1, H->o2, H->d_stats, H->out1, H->out2, n, T, cf.align, c.st));
    // motion_embed + PE of the four streams (mixermdm.py:722-732); seq = p*n + b
    RC(repack(c, H->mx, H->out1, NF2, H->sa.xp, 2, (int)nT));
    RC(repack(c, H->mx, H->out2, NF2, H->sb.xp, 2, (int)nT));
    StackRun r;
    r.nseq = 2 * n; r.T = T; r.ss = H->ss_mx; r.ss_ld = ss_ld_of(H->mx);
    r.sa_row0 = 0; r.sa_rows = 2 * n;          // cond_i1 | cond_i2
    r.ca_row0 = 2 * n; r.ca_rows = n;          // cond_I
    r.ffn_row0 = 2 * n; r.ffn_rows = n;        // FFN is conditioned on cond_I (influence.py:46)
    r.ca_mode = 2; r.kv_src = H->mI;
    const bool split = H->overlap && !H->prof.on && c.s == &H->sa && H->sb.h;      // the two Influence calls on two streams, each with its own scratch
    for (int p = 0; p < 2; ++p) {
        RC(embed(c, H->mx, H->sa.xp, p, c.s->h + (size_t)p * nT * Dm, n, T));
        RC(embed(c, H->mx, H->sb.xp, p, H->mI + (size_t)p * nT * Dm, n, T));
    }
    if (split) {
        // the two Influence calls (mixermdm.py:735-736: person 1, person 2) are independent: one per stream (64.7 -> 64.2 ms/step)
        Ctx c2{H, H->st2, &H->sb, c.g};
        StackRun r1 = r, r2 = r;
        r1.nseq = r2.nseq = n;
        r1.sa_rows = r2.sa_rows = n;
        r2.sa_row0 = n;
        r2.kv_src = H->mI + nT * Dm;
        HIPCHK(hipEventRecord(H->ev_fork2, c.st));
        HIPCHK(hipStreamWaitEvent(H->st2, H->ev_fork2, 0));
        RC(run_stack(c2, H->mx.st, c.s->h + nT * Dm, r2));
        RC(run_stack(c, H->mx.st, c.s->h, r1));
        HIPCHK(hipEventRecord(H->ev_join2, H->st2));
        HIPCHK(hipStreamWaitEvent(c.st, H->ev_join2, 0));
    } else {
        RC(run_stack(c, H->mx.st, c.s->h, r));
    }
    const int mode = cf.mixing_mode;
    // Influence.out + sigmoid (influence.py:124-125) as a GEMM with a sigmoid epilogue: N = 1 or 23 columns of a 64-wide MFMA tile --
    // wasteful per flop and still 8x faster than a wave-per-row dot-product kernel at 19 200 rows
    if (mode == 1 || mode == 3) {
        if (rag) RC(ROWOP(c, mmdm_mean_time_rag, c.s->h, H->hpool, 2 * n, c.g->seq_off, c.g->seq_len, Dm, c.st));
        else RC(ROWOP(c, mmdm_mean_time_f32, c.s->h, H->hpool, 2 * n, T, Dm, c.st));
        RC(linear(c, H->hpool, Dm, H->mx.out_w, Dm, H->mx.out_b, H->w23, H->nw, 2 * n, H->nw, Dm, MMDM_EPI_BIAS_SIGMOID));
    } else {
        RC(linear(c, c.s->h, Dm, H->mx.out_w, Dm, H->mx.out_b, H->w23, H->nw, (int)(2 * nT), H->nw, Dm, MMDM_EPI_BIAS_SIGMOID));
    }
    // history destinations come from the device-side descriptor: the launches below are identical whether or not (and where) a call
    // keeps history, so one captured graph serves all of them; the two copy kernels return at once on a null destination
    const int* lp = H->d_step + 1;
    if (rag) RC(mmdm_blend_cfg_rag(H->out1, H->out2, H->w23, mode, cf.use_force, cf.force_val, cf.cfg_scale, H->model_out, H->d_hist, lp, c.g->rg, c.st));
    else RC(mmdm_blend_cfg_dyn(H->out1, H->out2, H->w23, mode, cf.use_force, cf.force_val, cf.cfg_scale, H->model_out, H->d_hist, lp, B, T, c.st));
    RC(mmdm_hist_copy(H->out1, H->d_hist, 0, nT * NF2, lp, c.st));
    RC(mmdm_hist_copy(H->out2, H->d_hist, 1, nT * NF2, lp, c.st));
    (void)dyn_hist;
    return MMDM_OK;
}

// One full sampler step on the handle's state (x, x2, step index on the device).
int run_step(const Ctx& c) {
    mmdm_handle H = c.h;
    const int B = H->B, T = H->T, n = 2 * B;
    // the single-chain samplers run ONE stream of kernels: split the GEMMs' fractional last round (gemm_f32.hip; results are bit-identical);
    // scoped: the calling thread's stateless mmdm_linear_f32 calls keep the default rule afterwards
    TailScope tail_scope((H->cfg.single_only == 1 || H->cfg.single_only == 2) ? 10 : 0);
    if (H->cfg.single_only == 1) {
        if (H->d1.kind == 1) {
            RC(run_denoiser_mdm(c, H->d1, H->x, B, 1, NF, n, T, H->cond_cat, H->td1, H->o1, NF));
        } else {
            RC(cond_vectors(c, H->d1, H->txt_d1, H->se_d1, H->ss_d1, n));
            RC(run_denoiser(c, H->d1, false, H->x, B, 1, NF, n, T, H->ss_d1, ss_ld_of(H->d1), H->o1, NF));
        }
        // (ragged: one "item" of `rows` frames -- the combine is element-wise over a half of the CFG-doubled batch, padding rows included)
        if (c.rag()) RC(mmdm_cfg_ddim_f32(H->o1, H->d_coef, H->S, H->d_step, H->cfg.cfg_scale, H->x, H->px1, 1, c.g->rows, NF, c.st));
        else RC(mmdm_cfg_ddim_f32(H->o1, H->d_coef, H->S, H->d_step, H->cfg.cfg_scale, H->x, H->px1, B, T, NF, c.st));
        return mmdm_step_dec(H->d_step, H->d_step + 1, c.st);
    }
    if (H->cfg.single_only == 2) {   // stand-alone interaction denoiser, 4 CFG copies of x (cfg_sampler.py:70-71)
        const int n4 = 4 * B;
        RC(cond_vectors(c, H->d2, H->txt_d2, H->se_d2, H->ss_d2, 3 * n4));
        RC(run_denoiser(c, H->d2, true, H->x, B, 2, NF2, n4, T, H->ss_d2, ss_ld_of(H->d2), H->o2, NF2));
        RC(mmdm_cfg4_ddim_f32(H->o2, H->d_coef, H->S, H->d_step, H->cfg.cfg_scale, H->cfg.cfg_scale_interaction, H->cfg.cfg_scale_individual,
                              H->x, H->px1, B, T, NF2, c.st));
        return mmdm_step_dec(H->d_step, H->d_step + 1, c.st);
    }
    const bool dual = H->cfg.single_only == 3;
    // the individual model on both persons: in2IN blocks, the "dual_individual" variant of them, or MDMDenoiser
    auto model1 = [&](const Ctx& cc) -> int {
        if (H->d1.kind == 1) return run_denoiser_mdm(cc, H->d1, H->x, B, 2, NF2, n, T, H->cond_cat + 3 * H->cfg.text_dim, H->cond_w, H->o1, NF2);
        if (dual) return run_dual_individual(cc, H->d1, H->x, B, n, T, H->ss_d1, ss_ld_of(H->d1), H->o1);
        return run_denoiser(cc, H->d1, false, H->x, B, 2, NF2, n, T, H->ss_d1, ss_ld_of(H->d1), H->o1, NF2);
    };
    const float* xin2 = dual ? H->x : H->x2;       // DualMDM feeds the same x to both models (cfg_sampler.py:141-142)
    if (H->overlap && !H->prof.on) {
        // fork: denoiser2 (with its AdaLN-projection GEMM) on the auxiliary stream with its own scratch; denoiser1 and the mixer's
        // projections on the caller's stream (denoiser1 is the shorter model); join before the mixer
        Ctx c2{H, H->st2, &H->sb, c.g};
        HIPCHK(hipEventRecord(H->ev_fork, c.st));
        HIPCHK(hipStreamWaitEvent(H->st2, H->ev_fork, 0));
        RC(cond_vectors(c2, H->d2, H->txt_d2, H->se_d2, H->ss_d2, 3 * n));
        RC(run_denoiser(c2, H->d2, true, xin2, B, 2, NF2, n, T, H->ss_d2, ss_ld_of(H->d2), H->o2, NF2));
        if (H->d1.kind == 0) RC(cond_vectors(c, H->d1, H->txt_d1, H->se_d1, H->ss_d1, 2 * n));
        RC(model1(c));
        if (!dual) RC(cond_vectors(c, H->mx, H->txt_mx, H->se_mx, H->ss_mx, 3 * n));
        HIPCHK(hipEventRecord(H->ev_join, H->st2));
        HIPCHK(hipStreamWaitEvent(c.st, H->ev_join, 0));
    } else {
        if (H->d1.kind == 0) RC(cond_vectors(c, H->d1, H->txt_d1, H->se_d1, H->ss_d1, 2 * n));
        RC(cond_vectors(c, H->d2, H->txt_d2, H->se_d2, H->ss_d2, 3 * n));
        if (!dual) RC(cond_vectors(c, H->mx, H->txt_mx, H->se_mx, H->ss_mx, 3 * n));
        RC(model1(c));
        RC(run_denoiser(c, H->d2, true, xin2, B, 2, NF2, n, T, H->ss_d2, ss_ld_of(H->d2), H->o2, NF2));
    }
    if (dual) {
        RC(mmdm_dual_ddim_f32(H->o1, H->o2, H->d_coef, H->S, H->d_step, H->dual_w, H->cfg.cfg_scale_individual, H->cfg.cfg_scale_interaction,
                              H->x, H->px1, B, T, NF2, c.st));
        return mmdm_step_dec(H->d_step, H->d_step + 1, c.st);
    }
    RC(mixer_core(c, B, T, true));
    if (c.rag()) RC(mmdm_xstart_ddim_rag(H->model_out, H->d_stats, H->d_coef, H->S, H->d_step, H->x, H->x2, H->px1, H->px2, H->floor_ws,
                                         H->cfg.xstart_align, c.g->rg, c.st));
    else RC(mmdm_xstart_ddim_f32(H->model_out, H->d_stats, H->d_coef, H->S, H->d_step, H->x, H->x2, H->px1, H->px2, H->floor_ws,
                                 B, T, H->cfg.xstart_align, c.st));
    return mmdm_step_dec(H->d_step, H->d_step + 1, c.st);
}

// text embeddings of the three modules from a CFG-doubled cond [n, 8*td] (rows B.. are zero: cfg_sampler.py:45-46)
int text_all(const Ctx& c, const float* cond, int n) {
    mmdm_handle H = c.h;
    const int td = H->cfg.text_dim, td1 = H->td1, ldc = H->cond_w, base = 3 * td + 2 * td1;
    // denoiser1: person 1 <- ind_ind1, person 2 <- ind_ind2 (mixermdm.py:672-673); MDMDenoiser consumes its slices directly
    if (H->d1.kind == 0) {
        RC(text_rows(c, H->d1, cond, ldc, 3 * td, H->txt_d1, 0, n));
        RC(text_rows(c, H->d1, cond, ldc, 3 * td + td1, H->txt_d1, n, n));
    }
    // denoiser2 rows: [emb_individual1 | emb_individual2 | emb(interaction)] (in2in.py:415-417); InterGen shares one emb (intergen.py:270)
    const bool ig = H->cfg.model2_kind == 1;
    RC(text_rows(c, H->d2, cond, ldc, ig ? 0 : 1 * td, H->txt_d2, 0, n));
    RC(text_rows(c, H->d2, cond, ldc, ig ? 0 : 2 * td, H->txt_d2, n, n));
    RC(text_rows(c, H->d2, cond, ldc, 0, H->txt_d2, 2 * n, n));
    if (H->cfg.single_only == 3) return MMDM_OK;       // dual: cond is [n, 5*td], no mixer
    // mixer rows: [cond_i1 | cond_i2 | cond_I] (mixermdm.py:677-682)
    RC(text_rows(c, H->mx, cond, ldc, base + td, H->txt_mx, 0, n));
    RC(text_rows(c, H->mx, cond, ldc, base + 2 * td, H->txt_mx, n, n));
    RC(text_rows(c, H->mx, cond, ldc, base, H->txt_mx, 2 * n, n));
    return MMDM_OK;
}

// time_tab[i] = time_embed(pe[timestep_map[i]])  (utils.py:54-55) for all respaced steps
int build_time_tab(const Ctx& c, ModuleW& m) {
    mmdm_handle H = c.h;
    const int D = m.st.D, S = H->S;
    RC(mmdm_gather_rows(m.pe, H->d_tmap, H->tt_tmp, S, D, c.st));
    RC(linear(c, H->tt_tmp, D, m.t0_w, D, m.t0_b, H->tt_tmp2, D, S, D, D, MMDM_EPI_BIAS_SILU));
    return linear(c, H->tt_tmp2, D, m.t2_w, D, m.t2_b, m.time_tab, D, S, D, D);
}

size_t max2(size_t a, size_t b) { return a > b ? a : b; }

// Destroys a graph exec -- unless other handles share this handle's weights (mmdm_create_shared: several calls in flight).  In this runtime
// (ROCm 7.0 / 7.2) destroying an exec while ANOTHER handle's execs are alive made that handle's next hipGraphLaunch crash inside
// hip::Graph::UpdateStreams (seen with 2 and 4 handles at the real model sizes as soon as a graph cache evicted; waiting for the whole device
// first does not help, never destroying does): such execs are parked for the life of the process instead.  Once an exec has been parked,
// parked execs are alive for good -- so from then on EVERY exec of the process is parked rather than destroyed (the full GPU suite crashed in
// a later, unrelated handle's hipGraphLaunch after the shared-handle test had parked some execs and later tests destroyed theirs).
// What bounds the parked set: while parking is in force a graph cache does NOT evict (must_park() in mmdm_run: the cache grows to the
// number of distinct shapes its handle samples, every shape is captured once and re-used), so only mmdm_destroy parks -- the execs of handles
// that no longer exist, i.e. (handles ever destroyed while parking) x (distinct shapes each had sampled).  A process that never lets two
// handles share weights (bench.py's headline, one Sampler) never parks, evicts at graph_cap and frees its execs as before.
// Caller holds g_graph_mu exclusively.
std::vector<hipGraphExec_t> g_parked_execs;
bool must_park(mmdm_handle h) { return (h->wb && h->wb.use_count() > 1) || !g_parked_execs.empty(); }
void retire_exec(mmdm_handle h, hipGraphExec_t exec) {
    if (!exec) return;
    if (must_park(h)) g_parked_execs.push_back(exec);
    else (void)hipGraphExecDestroy(exec);
}

void drop_graphs(mmdm_handle h) {
    std::unique_lock<std::shared_mutex> lock(g_graph_mu);
    for (auto& g : h->graphs) {
        retire_exec(h, g.exec);
        if (g.done) (void)hipEventDestroy(g.done);
    }
    h->graphs.clear();
}

// profiling is switched off around set-up launches (they are not part of a step) and restored on every exit path
struct ProfPause {
    Prof& p; bool was;
    explicit ProfPause(Prof& pr) : p(pr), was(pr.on) { p.on = false; }
    ~ProfPause() { p.on = was; }
};

int push_hist(mmdm_handle h, hipStream_t st) { return mmdm_set_hist_desc(h->d_hist, h->hist, st); }

}  // namespace

// ------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------
extern "C" const char* mmdm_handle_error(mmdm_handle h) { return h ? h->err : "null handle"; }

// mmdm_create (parent == nullptr) and mmdm_create_shared (parent = the handle whose weights the new one borrows)
static int create_impl(const mmdm_config* cfg, mmdm_handle parent, mmdm_handle* out) {
    if (!cfg || !out) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create: null argument");
    if (cfg->nfeats != NF) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_create: nfeats must be 262");
    const int so = cfg->single_only;
    if (so < 0 || so > 3) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create: single_only must be 0, 1, 2 or 3");
    const bool has_d1 = so != 2, has_d2 = so != 1, has_mx = so == 0, two_models = so == 0 || so == 3;
    if (cfg->model1_kind < 0 || cfg->model1_kind > 1) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create: model1_kind must be 0 (in2IN individual) or 1 (MDM)");
    const bool mdm = has_d1 && cfg->model1_kind == 1;
    if (mdm && so == 3) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create: the dual sampler composes two in2IN denoisers (model1_kind must be 0)");
    // denoiser1 may have its own sizes (MODEL1 and MODEL2 are separate configs: src/models/mixermdm.py:32-40); 0 = same as denoiser2
    const int D = cfg->d_latent, F = cfg->d_ff;
    const int D1 = cfg->d1_latent ? cfg->d1_latent : D, F1 = cfg->d1_ff ? cfg->d1_ff : F;
    const int L1 = cfg->d1_layers ? cfg->d1_layers : cfg->d_layers, H1 = cfg->d1_heads ? cfg->d1_heads : cfg->d_heads;
    if (D <= 0 || cfg->d_heads <= 0 || D % cfg->d_heads || D % 4 || F <= 0 || F % 4 || cfg->d_layers <= 0 ||
        D1 <= 0 || H1 <= 0 || D1 % H1 || D1 % 4 || F1 <= 0 || F1 % 4 || L1 <= 0)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create: bad denoiser dims");
    if (has_mx && (cfg->m_latent <= 0 || cfg->m_heads <= 0 || cfg->m_latent % cfg->m_heads || cfg->m_latent % 4 || cfg->m_ff <= 0 || cfg->m_ff % 4 || cfg->m_layers <= 0))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create: bad mixer dims");
    if (has_mx && (cfg->mixing_mode < 1 || cfg->mixing_mode > 4)) return mmdm_set_error(MMDM_ERR_ARG, "Mode not recognized");
    if (cfg->max_batch <= 0 || cfg->max_frames <= 0 || cfg->text_dim <= 0 || cfg->text_dim % 4)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create: bad max_batch / max_frames / text_dim");
    if (cfg->precision < 0 || cfg->precision > 3)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create: precision must be 0 (fp32 MFMA), 1 (bf16 GEMM operands), 2 (fp32 by exact bf16 operand splitting) or 3 (bf16 + fp8 QKV/FFN operands)");
    if (cfg->precision == 3 && ((D % 64) || (F % 64) || (D1 % 64) || (F1 % 64) || (has_mx && ((cfg->m_latent % 64) || (cfg->m_ff % 64)))))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_create: the fp8 path needs latent and ff sizes that are multiples of 64");
    if (cfg->precision >= 1 && ((D % 32) || (F % 32) || (D1 % 32) || (F1 % 32) || (has_mx && ((cfg->m_latent % 32) || (cfg->m_ff % 32)))))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_create: the bf16 path needs latent and ff sizes that are multiples of 32");
    if (cfg->precision >= 1 && mdm) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_create: the bf16 / fp32-split paths do not cover MDMDenoiser");
    RC(mmdm_kernels_init());
    mmdm_handle h = new mmdm_handle_s();
    h->cfg = *cfg;
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return mmdm_set_error(MMDM_ERR_HIP, "mmdm_create: hipGetDevice failed"); }
    if (parent && parent->device != h->device) { delete h; return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create_shared: the parent handle lives on device %d, the current device is %d", parent->device, h->device); }
    const mmdm_config& c = h->cfg;
    h->nw = (c.mixing_mode >= 3) ? 23 : 1;
    int rc = MMDM_OK;
    auto fail = [&](int code) { mmdm_destroy(h); return code; };
    const int B = c.max_batch, T = c.max_frames, n = (so == 2 ? 4 : 2) * B, td = c.text_dim;
    const int Dm = has_mx ? c.m_latent : 4, Fm = has_mx ? c.m_ff : 4;
    h->td1 = mdm ? D1 : td;                                  // MDMDenoiser adds its cond slice to the timestep embedding: latent-sized (mdm.py:279)
    h->cond_w = so == 0 ? 6 * td + 2 * h->td1 : so == 1 ? h->td1 : so == 2 ? 3 * td : 5 * td;
    if (parent) {
        // borrow the parent's weights: the module descriptors are tables of pointers into the shared weight block (incl. the low-precision
        // twins and their layout flags as mmdm_prepare left them); everything that depends on a call or a schedule is this handle's own
        h->wb = parent->wb;
        h->shared_child = true;
        h->d1 = parent->d1; h->d2 = parent->d2; h->mx = parent->mx;
        h->d1.time_tab = h->d2.time_tab = h->mx.time_tab = nullptr;
        h->d1.pe_r = h->d2.pe_r = h->mx.pe_r = nullptr;
        h->d_stats = parent->d_stats; h->stats_set = parent->stats_set;
        h->prepared = true;
    } else {
        h->wb = std::make_shared<mmdm_weight_block>();
        h->wb->device = h->device;
        h->alloc_weights = true;
        if (has_d1) {
            rc = mdm ? build_module_mdm(h, h->d1, "denoiser1.", D1, F1, L1, H1)
                     : build_module(h, h->d1, "denoiser1.", "denoiser1.", D1, F1, L1, H1, false, true, "denoiser1.out.linear", NF);
            if (rc) return fail(rc);
        }
        if (has_d2 && (rc = build_module(h, h->d2, "denoiser2.", "denoiser2.", D, F, c.d_layers, c.d_heads, true, false, "denoiser2.out.linear", NF))) return fail(rc);
        if (has_mx && (rc = build_module(h, h->mx, "", "influence.", Dm, Fm, c.m_layers, c.m_heads, true, false, "influence.out", h->nw))) return fail(rc);
        if ((rc = dalloc(h, &h->d_stats, 4 * NF))) return fail(rc);
        h->alloc_weights = false;
    }
    // per-schedule timestep-embedding tables: this handle's own (a shared handle may run another sampling strategy than its parent)
    if (has_d1 && (rc = dalloc(h, &h->d1.time_tab, (size_t)h->Smax * D1))) return fail(rc);
    if (has_d2 && (rc = dalloc(h, &h->d2.time_tab, (size_t)h->Smax * D))) return fail(rc);
    if (has_mx && (rc = dalloc(h, &h->mx.time_tab, (size_t)h->Smax * Dm))) return fail(rc);
    const int npers = so == 1 ? 1 : 2;
    const size_t R = (size_t)npers * n * (T + 1);            // + 1: MDMDenoiser's conditioning token
    const size_t Dx = max2(max2(has_d1 ? D1 : 4, has_d2 ? D : 4), Dm), Fx = max2(max2(has_d1 ? F1 : 4, has_d2 ? F : 4), Fm);
    for (Scratch* sc : {&h->sa, &h->sb}) {
        if (sc == &h->sb && !two_models) break;
        const size_t d = sc == &h->sa ? Dx : (size_t)D, f = sc == &h->sa ? Fx : (size_t)F;
        // GEMM-operand buffers (xn, att, f1): fp32, or the two fp16 planes of the fp32-split mode -- 4 bytes per element either way
        const size_t opx = c.precision == 2 ? MMDM_SPLIT_NPL : 2;       // in half-floats
        if ((rc = dalloc(h, &sc->h, R * d)) || (rc = dalloc(h, &sc->xn, R * d * opx / 2)) || (rc = dalloc(h, &sc->att, R * d * opx / 2)) ||
            (rc = dalloc(h, &sc->qkv, R * 3 * d)) || (rc = dalloc(h, &sc->kv, R * 2 * d)) || (rc = dalloc(h, &sc->f1, R * f * opx / 2)) ||
            (rc = dalloc(h, &sc->xp, (size_t)2 * n * T * (NFS > NFP ? NFS : NFP))))
            return fail(rc);
        if (c.precision >= 1) {
            const size_t npl = c.precision == 2 ? 3 : 1;
            float *q1 = nullptr, *q2 = nullptr;
            // bf16 [npl][R][2d] (Q|K) and [npl][R][d] (cross-attention K); the one-plane modes also keep V there ([R][3d], [R][2d]): P.V on the bf16 cores
            const size_t vx = npl == 1 ? 1 : 0;
            if ((rc = dalloc(h, &q1, npl * R * d + vx * R * d / 2 + 1)) || (rc = dalloc(h, &q2, npl * R * d / 2 + vx * R * d / 2 + 1))) return fail(rc);
            sc->qk = q1; sc->kvp = q2;
            if (c.precision == 3 && (rc = dalloc(h, &sc->xs, R))) return fail(rc);
        }
    }
    if (hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join2, hipEventDisableTiming) != hipSuccess)
        return fail(mmdm_set_error(MMDM_ERR_HIP, "mmdm_create: stream/event creation failed"));
    h->overlap = getenv("MMDM_NO_OVERLAP") == nullptr;
    if (g_serialize_handles.load() < 0) { const char* v = getenv("MMDM_SERIALIZE_HANDLES"); g_serialize_handles.store(v && strcmp(v, "0") != 0 ? 1 : 0); }
    auto env_on = [](const char* k) { const char* v = getenv(k); return v != nullptr && strcmp(v, "0") != 0; };
    if (parent) {       // the switches that decide the FORMAT of the shared weights (and the kernels that read them) are the parent's
        h->force_qkp = parent->force_qkp; h->no_qkp = parent->no_qkp; h->no_pvb = parent->no_pvb; h->no_pack = parent->no_pack; h->no_split_embed = parent->no_split_embed;
        h->no_split_cond = parent->no_split_cond;
    } else {
        h->force_qkp = env_on("MMDM_QKP"); h->no_qkp = env_on("MMDM_NO_QKP"); h->no_pvb = env_on("MMDM_NO_BF16_PV");
        h->no_pack = env_on("MMDM_NO_PACK") || env_on("MMDM_SPLIT_NO_PACK");
        h->no_split_embed = env_on("MMDM_NO_SPLIT_EMBED");
        h->no_split_cond = env_on("MMDM_NO_SPLIT_COND");
    }
    const size_t PT = (size_t)n * T * (so == 1 ? NF : NF2);
    const size_t PB = (size_t)B * T * (so == 1 ? NF : NF2);
    if ((rc = dalloc(h, &h->x, PB)) || (rc = dalloc(h, &h->px1, PB))) return fail(rc);
    if (has_d1) {
        if ((rc = dalloc(h, &h->o1, PT))) return fail(rc);
        if (!mdm && ((rc = dalloc(h, &h->txt_d1, (size_t)npers * n * D1)) || (rc = dalloc(h, &h->se_d1, (size_t)npers * n * D1)) ||
                     (rc = dalloc(h, &h->ss_d1, (size_t)npers * n * ss_ld_of(h->d1)))))
            return fail(rc);
    }
    if (has_d2) {
        if ((rc = dalloc(h, &h->o2, PT)) || (rc = dalloc(h, &h->txt_d2, (size_t)3 * n * D)) || (rc = dalloc(h, &h->se_d2, (size_t)3 * n * D)) ||
            (rc = dalloc(h, &h->ss_d2, (size_t)3 * n * ss_ld_of(h->d2))))
            return fail(rc);
    }
    if (has_mx) {
        if ((rc = dalloc(h, &h->mI, R * Dm)) || (rc = dalloc(h, &h->out1, PT)) || (rc = dalloc(h, &h->out2, PT)) ||
            (rc = dalloc(h, &h->w23, (size_t)2 * n * T * 23)) || (rc = dalloc(h, &h->hpool, (size_t)2 * n * Dm)) ||
            (rc = dalloc(h, &h->model_out, PB)) || (rc = dalloc(h, &h->x2, PB)) || (rc = dalloc(h, &h->px2, PB)) ||
            (rc = dalloc(h, &h->floor_ws, (size_t)2 * B)) ||
            (rc = dalloc(h, &h->txt_mx, (size_t)3 * n * Dm)) || (rc = dalloc(h, &h->se_mx, (size_t)3 * n * Dm)) ||
            (rc = dalloc(h, &h->ss_mx, (size_t)3 * n * ss_ld_of(h->mx))))
            return fail(rc);
    }
    if ((rc = dalloc(h, &h->cond_cat, (size_t)n * h->cond_w))) return fail(rc);
    if ((rc = dalloc(h, &h->tt_tmp, (size_t)h->Smax * Dx)) || (rc = dalloc(h, &h->tt_tmp2, (size_t)h->Smax * Dx)) || (rc = dalloc(h, &h->tt_stash, 3 * Dx))) return fail(rc);
    if ((rc = dalloc(h, &h->d_coef, (size_t)4 * h->Smax)) || (rc = dalloc(h, &h->dual_w, h->Smax))) return fail(rc);
    float* tmp = nullptr;
    if ((rc = dalloc(h, &tmp, h->Smax))) return fail(rc);
    h->d_tmap = reinterpret_cast<int*>(tmp);
    if ((rc = dalloc(h, &tmp, 4))) return fail(rc);
    h->d_step = reinterpret_cast<int*>(tmp);
    if ((rc = dalloc(h, &tmp, (sizeof(mmdm_hist_desc) + 3) / 4))) return fail(rc);
    h->d_hist = reinterpret_cast<mmdm_hist_desc*>(tmp);
    if (hipMemcpy(h->d_hist, &h->hist, sizeof(mmdm_hist_desc), hipMemcpyHostToDevice) != hipSuccess)
        return fail(mmdm_set_error(MMDM_ERR_HIP, "mmdm_create: history descriptor upload failed"));
    if (const char* e = getenv("MMDM_GRAPH_CACHE")) { long v = atol(e); if (v >= 1 && v <= 64) h->graph_cap = (size_t)v; }
    // ragged calls (mmdm_begin_ragged): row maps for up to 4 groups of max_batch * max_frames rows, and per module the PE rows of a group
    if (so <= 1 && !mdm) {
        const size_t cap = (size_t)B * T, nb = (size_t)(B < MMDM_RAG_MAX_ITEMS ? B : MMDM_RAG_MAX_ITEMS), G = 4;
        if ((rc = dalloc(h, &tmp, 3 * nb + 2 * cap + G * cap + 2 * G * nb))) return fail(rc);
        h->d_rag = reinterpret_cast<int*>(tmp);
        h->d_item_off = h->d_rag; h->d_item_len = h->d_item_off + nb; h->d_row_item = h->d_item_len + nb; h->d_row_pos = h->d_row_item + cap;
        h->d_row_seq = h->d_row_pos + cap; h->d_seq_off = h->d_row_seq + G * cap; h->d_seq_len = h->d_seq_off + G * nb; h->d_item_order = h->d_seq_len + G * nb;
        if (has_d1 && (rc = dalloc(h, &h->d1.pe_r, cap * D1))) return fail(rc);
        if (has_d2 && (rc = dalloc(h, &h->d2.pe_r, cap * D))) return fail(rc);
        if (has_mx && (rc = dalloc(h, &h->mx.pe_r, cap * Dm))) return fail(rc);
        if (const char* e = getenv("MMDM_RAG_BUCKET")) { long v = atol(e); if (v >= 1 && v <= 4096) h->rag_bucket = (int)v; }
    }
    *out = h;
    return MMDM_OK;
}

extern "C" int mmdm_create(const mmdm_config* cfg, mmdm_handle* out) { return create_impl(cfg, nullptr, out); }

extern "C" int mmdm_create_shared(mmdm_handle parent, int max_batch, int max_frames, mmdm_handle* out) {
    if (!parent || !out) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_create_shared: null argument");
    if (!parent->prepared) return herr(parent, mmdm_set_error(MMDM_ERR_STATE, "mmdm_create_shared: load the weights and call mmdm_prepare on the parent first"));
    mmdm_config c = parent->cfg;
    c.max_batch = max_batch; c.max_frames = max_frames;
    return create_impl(&c, parent, out);
}

extern "C" void mmdm_destroy(mmdm_handle h) {
    if (!h) return;
    // the handle's streams, events and buffers live on the device it was created on, which need not be the caller's current one
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != h->device) (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    drop_graphs(h);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_fork2) (void)hipEventDestroy(h->ev_fork2);
    if (h->ev_join2) (void)hipEventDestroy(h->ev_join2);
    if (h->st2) (void)hipStreamDestroy(h->st2);
    for (int k = 0; k < Prof::NCLS; ++k)
        for (hipEvent_t e : h->prof.ev[k]) (void)hipEventDestroy(e);
    for (void* p : h->allocs) (void)hipFree(p);
    if (cur >= 0 && cur != h->device) (void)hipSetDevice(cur);
    delete h;
}

extern "C" int mmdm_set_weight(mmdm_handle h, const char* name, const float* src, int64_t rows, int64_t cols, void* stream) {
    if (!h || !name) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_weight: null argument");
    if (h->shared_child) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_set_weight(%s): this handle borrows its weights (mmdm_create_shared); set them on the parent", name));
    auto it = h->slots.find(name);
    if (it == h->slots.end()) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "Unexpected key in state_dict: \"%s\"", name));
    Slot& s = it->second;
    if (!s.dst) { s.set = true; return MMDM_OK; }   // accepted and ignored
    if (!src) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_weight(%s): null source", name));
    if (rows != s.rows || cols != s.cols)
        return herr(h, mmdm_set_error(MMDM_ERR_ARG, "size mismatch for %s: got [%lld, %lld], expected [%lld, %lld]", name, (long long)rows,
                                      (long long)cols, (long long)s.rows, (long long)s.cols));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpy2DAsync(s.dst, s.ld * sizeof(float), src, cols * sizeof(float), cols * sizeof(float), rows, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return herr(h, mmdm_set_error(MMDM_ERR_HIP, "mmdm_set_weight(%s): %s", name, hipGetErrorString(e)));
    s.set = true;
    h->prepared = false;
    return MMDM_OK;
}

extern "C" int mmdm_set_norm_stats(mmdm_handle h, const float* stats_host) {
    if (!h || !stats_host) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_norm_stats: null argument");
    if (h->shared_child) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_set_norm_stats: this handle borrows its weights and statistics (mmdm_create_shared); set them on the parent"));
    hipError_t e = hipMemcpy(h->d_stats, stats_host, 4 * NF * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) return herr(h, mmdm_set_error(MMDM_ERR_HIP, "mmdm_set_norm_stats: %s", hipGetErrorString(e)));
    h->stats_set = true;
    return MMDM_OK;
}

extern "C" int mmdm_prepare(mmdm_handle h) {
    if (!h) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_prepare: null handle");
    if (h->shared_child) return MMDM_OK;               // prepared through the parent (mmdm_create_shared requires it)
    std::string missing;
    int nmiss = 0;
    for (auto& kv : h->slots)
        if (kv.second.required && !kv.second.set) {
            if (nmiss < 4) missing += (nmiss ? ", " : "") + kv.first;
            ++nmiss;
        }
    if (nmiss) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "Missing key(s) in state_dict (%d): %s%s", nmiss, missing.c_str(), nmiss > 4 ? ", ..." : ""));
    if (h->cfg.single_only == 0 && !h->stats_set) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_prepare: normaliser stats not set"));
    if (h->cfg.precision >= 1) {
        const bool split = h->cfg.precision == 2;
        // Low-precision weights of a stack whose sizes the packed kernels cover (fp32-split: every N and K a multiple of 128; bf16 / fp8: of 256;
        // slices on 32-row boundaries) are stored in fragment order: converted into a scratch buffer, then permuted into place (same size)
        const bool no_pack = h->no_pack;
        void* tmp = nullptr;
        size_t tmp_elems = 0;
        for (ModuleW* m : {&h->d1, &h->d2, &h->mx}) {
            StackW& st = m->st;
            const int gran = split ? 128 : 256;             // the packed kernels' N / K granularity (gemm_split.hip: 128 x 128 tiles; gemm_bf16.hip: 128 x 256, K step 128 bytes)
            st.w_packed = !no_pack && !st.layers_b.empty() && st.D % gran == 0 && st.F % gran == 0;
            if (st.w_packed) tmp_elems = std::max(tmp_elems, (size_t)(split ? MMDM_SPLIT_NPL : 1) * std::max(3 * st.D, st.F) * st.D);
        }
        if (tmp_elems) HIPCHK(hipMalloc(&tmp, tmp_elems * 2));
        struct TmpFree { void* p; ~TmpFree() { if (p) (void)hipFree(p); } } tmp_free{tmp};
        bool pack_now = false;
        auto conv = [&](const float* src, void* dst, int64_t n, int64_t K = 0) -> int {
            if (!split) {
                if (!pack_now) return mmdm_f32_to_bf16(src, dst, n, nullptr);
                if (int rc = mmdm_f32_to_bf16(src, tmp, n, nullptr)) return rc;
                return mmdm_pack_weight_frag(tmp, 2 * K, dst, (int)(n / K), (int)(2 * K), nullptr);
            }
            if (!pack_now) return mmdm_f32_split(src, dst, n, n, nullptr);
            if (int rc = mmdm_f32_split(src, tmp, n, n, nullptr)) return rc;
            return mmdm_split_pack_weight(tmp, (int)K, n, dst, n, (int)(n / K), (int)K, nullptr);
        };
        for (ModuleW* m : {&h->d1, &h->d2, &h->mx}) {
            StackW& st = m->st;
            pack_now = st.w_packed;
            for (size_t i = 0; i < st.layers_b.size(); ++i) {
                const LayerW& lw = st.layers[i];
                LayerWB& lb = st.layers_b[i];
                const int64_t D = st.D, F = st.F;
                int rc = MMDM_OK;
                if (h->cfg.precision == 3) {          // per-output-channel e4m3 for QKV / cross-attention inputs / FFN, bf16 for the output projections
                    // the bf16 output projections of this mode are packed like the bf16 mode's
                    auto q8 = [&](const float* src, void* dst, float* sc, int rows, int cols) { return mmdm_quantize_rows_fp8(src, cols, dst, cols, sc, rows, cols, nullptr); };
                    // quantise into the scratch buffer, then permute into MFMA fragment order (the packed kernel takes W straight from global memory)
                    auto q8p = [&](const float* src, void* dst, float* sc, int rows, int cols) {
                        int r2 = mmdm_quantize_rows_fp8(src, cols, tmp, cols, sc, rows, cols, nullptr);
                        return r2 ? r2 : mmdm_pack_weight_frag(tmp, cols, dst, rows, cols, nullptr);
                    };
                    // Round 3, with the block-scaled 64-deep fp8 MFMA in both kernels (tools/gemm_fp8_bench.py, M = 19 200): the packed kernel now wins
                    // on the projections (QKV 1203 vs 988 TFLOP/s) and on K = 2048 (FFN-2 917 vs 685); FFN-1 stayed staged until round 5 (1048 vs 988)
                    rc = pack_now ? q8p(lw.sa_in_w, lb.sa_in_8, lb.sa_in_s, (int)(3 * D), (int)D) : q8(lw.sa_in_w, lb.sa_in_8, lb.sa_in_s, (int)(3 * D), (int)D);
                    if (!rc) rc = conv(lw.sa_out_w, lb.sa_out_w, D * D, D);
                    // (round 5: with the epilogue's de-quantisation operands in LDS the packed kernel wins on FFN-1 too -- 67.7 vs 76.7 us at M = 19 200)
                    if (!rc) rc = pack_now ? q8p(lw.f1_w, lb.f1_8, lb.f1_s, (int)F, (int)D) : q8(lw.f1_w, lb.f1_8, lb.f1_s, (int)F, (int)D);
                    if (!rc && pack_now && F >= 2048) {       // the one fp8 GEMM with K = 2048 (16 steps of the packed kernel): 1039 vs 901 TFLOP/s
                        rc = mmdm_quantize_rows_fp8(lw.f2_w, (int)F, tmp, (int)F, lb.f2_s, (int)D, (int)F, nullptr);
                        if (!rc) rc = mmdm_pack_weight_frag(tmp, F, lb.f2_8, (int)D, (int)F, nullptr);
                    } else if (!rc) rc = q8(lw.f2_w, lb.f2_8, lb.f2_s, (int)D, (int)F);
                    if (!rc && st.has_ca) rc = pack_now ? q8p(lw.ca_in_w, lb.ca_in_8, lb.ca_in_s, (int)(3 * D), (int)D) : q8(lw.ca_in_w, lb.ca_in_8, lb.ca_in_s, (int)(3 * D), (int)D);
                    if (!rc && st.has_ca) rc = conv(lw.ca_out_w, lb.ca_out_w, D * D, D);
                    if (rc) return herr(h, rc);
                    continue;
                }
                rc = conv(lw.sa_in_w, lb.sa_in_w, 3 * D * D, D);
                if (!rc) rc = conv(lw.sa_out_w, lb.sa_out_w, D * D, D);
                if (!rc) rc = conv(lw.f1_w, lb.f1_w, F * D, D);
                if (!rc) rc = conv(lw.f2_w, lb.f2_w, D * F, F);
                if (!rc && st.has_ca) rc = conv(lw.ca_in_w, lb.ca_in_w, 3 * D * D, D);
                if (!rc && st.has_ca) rc = conv(lw.ca_out_w, lb.ca_out_w, D * D, D);
                if (rc) return herr(h, rc);
            }
        }
        // motion_embed [D, NFP] fp32 -> zero-padded [D, NFS] -> two fp16 planes (-> fragment order unless MMDM_NO_PACK)
        for (ModuleW* m : {&h->d1, &h->d2, &h->mx}) {
            if (h->no_split_embed) m->me_s = nullptr;        // the embedding stays on the fp32 MFMA kernel (A/B switch)
            if (!m->me_s) continue;
            const int64_t D = m->st.D, n = D * NFS;
            float* w32 = nullptr; void* pl = nullptr;
            HIPCHK(hipMalloc(&w32, n * 4));
            struct F1 { void* p; ~F1() { (void)hipFree(p); } } f1{w32};
            HIPCHK(hipMalloc(&pl, n * 4));
            struct F2 { void* p; ~F2() { (void)hipFree(p); } } f2{pl};
            HIPCHK(hipMemsetAsync(w32, 0, n * 4, nullptr));
            HIPCHK(hipMemcpy2DAsync(w32, NFS * sizeof(float), m->me_w, NFP * sizeof(float), NFP * sizeof(float), D, hipMemcpyDeviceToDevice, nullptr));
            int rc = mmdm_f32_split(w32, no_pack ? m->me_s : pl, n, n, nullptr);
            if (!rc && !no_pack) rc = mmdm_split_pack_weight(pl, NFS, n, m->me_s, n, (int)D, NFS, nullptr);
            if (rc) return herr(h, rc);
            HIPCHK(hipDeviceSynchronize());
        }
        // the AdaLN projection matrix of each stack [L*n_ada*2D, D] -> two fp16 planes in fragment order (cond_vectors: a weight-streaming GEMM of
        // <= 6B rows, MFMA-bound on the fp32 kernel's 128-row tile, HBM-bound on the fp32-split one)
        for (ModuleW* m : {&h->d1, &h->d2, &h->mx}) {
            StackW& st = m->st;
            if (h->no_split_cond) st.ada_s = nullptr;       // the projections stay on the fp32 MFMA kernel (A/B switch)
            if (!st.ada_s) continue;
            const int64_t N = (int64_t)st.L * st.n_ada * 2 * st.D, n = N * st.D;
            void* pl = nullptr;
            if (!no_pack) HIPCHK(hipMalloc(&pl, n * 4));
            struct F3 { void* p; ~F3() { if (p) (void)hipFree(p); } } f3{pl};
            int rc = mmdm_f32_split(st.ada_w, no_pack ? st.ada_s : pl, n, n, nullptr);
            if (!rc && !no_pack) rc = mmdm_split_pack_weight(pl, st.D, n, st.ada_s, n, (int)N, st.D, nullptr);
            if (rc) return herr(h, rc);
            HIPCHK(hipDeviceSynchronize());
        }
        HIPCHK(hipDeviceSynchronize());
    }
    h->prepared = true;
    return MMDM_OK;
}

extern "C" int mmdm_set_schedule(mmdm_handle h, const int* timestep_map, const float* coef, int S, void* stream) {
    if (!h || !timestep_map || !coef) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_schedule: null argument");
    if (S <= 0 || S > h->Smax) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_schedule: S=%d outside [1, %d]", S, h->Smax));
    if (!h->prepared) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_set_schedule: call mmdm_prepare first"));
    for (int i = 0; i < S; ++i)
        if (timestep_map[i] < 0 || timestep_map[i] >= 5000) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_schedule: timestep %d out of the pe table", timestep_map[i]));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipMemcpyAsync(h->d_tmap, timestep_map, S * sizeof(int), hipMemcpyHostToDevice, st));
    for (int k = 0; k < 4; ++k)
        HIPCHK(hipMemcpyAsync(h->d_coef + (size_t)k * S, coef + (size_t)k * S, S * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // host buffers may be transient
    h->S = S;
    h->h_tmap0 = timestep_map[0];
    h->dual_w_set = false;
    // cached step graphs stay valid: S is part of their key, the tables are device data at fixed addresses
    Ctx c{h, st, &h->sa};
    ProfPause pause(h->prof);
    int rc = h->cfg.single_only != 2 ? build_time_tab(c, h->d1) : MMDM_OK;
    if (!rc && h->cfg.single_only != 1) rc = build_time_tab(c, h->d2);
    if (!rc && h->cfg.single_only == 0) rc = build_time_tab(c, h->mx);
    h->begun = false;
    return herr(h, rc);
}

extern "C" int mmdm_set_dual_weights(mmdm_handle h, const float* w_host, int S) {
    if (!h || !w_host) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_dual_weights: null argument");
    if (h->cfg.single_only != 3) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_set_dual_weights: the handle is not a dual sampler (single_only != 3)"));
    if (S != h->S || S <= 0) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_dual_weights: S=%d does not match the schedule (%d steps)", S, h->S));
    HIPCHK(hipMemcpy(h->dual_w, w_host, (size_t)S * sizeof(float), hipMemcpyHostToDevice));
    h->dual_w_set = true;
    return MMDM_OK;
}

// mmdm_begin (lens == nullptr: B items of T frames) and mmdm_begin_ragged (lens = B host ints, x_T = the items' frames back to back)
static int begin_impl(mmdm_handle h, const float* cond, const float* x_T, int B, int T, const int* lens, void* stream) {
    if (!h || !cond || !x_T) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_begin: null argument");
    if (!h->prepared || h->S == 0) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_begin: prepare() and set_schedule() first"));
    Geom g;
    if (lens) {
        if (h->cfg.single_only > 1 || h->d1.kind == 1 || !h->d_rag)
            return herr(h, mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_begin_ragged: ragged batches cover the two-chain MixerMDM sampler and the single-person sampler over in2IN / InterGen denoisers"));
        const StackW* sts[3] = {h->cfg.single_only != 2 ? &h->d1.st : nullptr, h->cfg.single_only != 1 ? &h->d2.st : nullptr, h->cfg.single_only == 0 ? &h->mx.st : nullptr};
        for (const StackW* w : sts)
            if (w && w->D / w->H != 64 && w->D / w->H != 128)
                return herr(h, mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_begin_ragged: head size %d (the ragged attention kernels cover 64 and 128)", w->D / w->H));
        if (B <= 0 || B > h->cfg.max_batch || B > MMDM_RAG_MAX_ITEMS)
            return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_begin_ragged: B=%d outside [1, min(max_batch=%d, %d)]", B, h->cfg.max_batch, MMDM_RAG_MAX_ITEMS));
        long sum = 0; int mx = 0; double tt1 = 0;
        for (int b = 0; b < B; ++b) {
            if (lens[b] <= 0 || lens[b] > h->cfg.max_frames)
                return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_begin_ragged: item %d has %d frames (1 .. max_frames=%d)", b, lens[b], h->cfg.max_frames));
            sum += lens[b]; mx = lens[b] > mx ? lens[b] : mx; tt1 += (double)lens[b] * (lens[b] + 1);
        }
        const long cap = (long)h->cfg.max_batch * h->cfg.max_frames;
        if (sum > cap) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_begin_ragged: %ld frames in all exceed the workspace (max_batch x max_frames = %ld)", sum, cap));
        long rows = (sum + h->rag_bucket - 1) / h->rag_bucket * h->rag_bucket;
        if (rows > cap) rows = cap;
        T = mx;
        g.rag = true; g.rows = (int)rows; g.real_rows = (int)sum; g.tt1 = tt1;
        g.seq_off = h->d_seq_off; g.seq_len = h->d_seq_len; g.row_seq = h->d_row_seq; g.item_order = h->d_item_order;
        g.rg = mmdm_rag{h->d_row_item, h->d_row_pos, h->d_item_off, h->d_item_len, B, (int)rows};
    } else {
        if (B <= 0 || B > h->cfg.max_batch || T <= 0 || T > h->cfg.max_frames)
            return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_begin: B=%d T=%d exceed the handle's max_batch=%d / max_frames=%d", B, T, h->cfg.max_batch, h->cfg.max_frames));
        g.rows = B * T; g.real_rows = B * T; g.tt1 = (double)B * T * (T + 1);
    }
    g.B = B; g.T = T;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ctx c{h, st, &h->sa};
    const int n = 2 * B, td = h->cfg.text_dim;
    // frames handed over / frame rows of the chains' buffers (ragged: the padding rows of the group are zeroed -- they are embedded and
    // normalised like any row, never read by a real one)
    const size_t fr = (size_t)g.real_rows, frp = (size_t)g.rows;
    auto load_x = [&](float* dst, int width) -> int {
        HIPCHK(hipMemcpyAsync(dst, x_T, fr * width * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (frp > fr) HIPCHK(hipMemsetAsync(dst + fr * width, 0, (frp - fr) * width * sizeof(float), st));
        return MMDM_OK;
    };
    if (g.rag) {
        RC(herr(h, mmdm_rag_setup(lens, B, g.rows, 4, h->d_item_off, h->d_item_len, h->d_row_item, h->d_row_pos, h->d_row_seq, h->d_seq_off, h->d_seq_len, h->d_item_order, st)));
        for (ModuleW* m : {&h->d1, &h->d2, &h->mx})
            if (m->pe_r && m->pe) RC(herr(h, mmdm_gather_rows(m->pe, h->d_row_pos, m->pe_r, g.rows, m->st.D, st)));
    }
    if (h->cfg.single_only == 3 && !h->dual_w_set) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_begin: call mmdm_set_dual_weights after mmdm_set_schedule"));
    ProfPause pause(h->prof);
    int rc = MMDM_OK;
    if (h->cfg.single_only == 2) {
        // 4 CFG copies of cond [B, 3*td]: full | interaction only (first td columns) | individuals only (columns td..) | zeros
        const int n4 = 4 * B, ldc = 3 * td;
        HIPCHK(hipMemsetAsync(h->cond_cat, 0, (size_t)n4 * ldc * sizeof(float), st));
        HIPCHK(hipMemcpyAsync(h->cond_cat, cond, (size_t)B * ldc * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpy2DAsync(h->cond_cat + (size_t)B * ldc, ldc * sizeof(float), cond, ldc * sizeof(float), td * sizeof(float), B, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpy2DAsync(h->cond_cat + (size_t)2 * B * ldc + td, ldc * sizeof(float), cond + td, ldc * sizeof(float), 2 * td * sizeof(float), B,
                                hipMemcpyDeviceToDevice, st));
        const bool ig = h->cfg.model2_kind == 1;
        rc = text_rows(c, h->d2, h->cond_cat, ldc, ig ? 0 : td, h->txt_d2, 0, n4);
        if (!rc) rc = text_rows(c, h->d2, h->cond_cat, ldc, ig ? 0 : 2 * td, h->txt_d2, n4, n4);
        if (!rc) rc = text_rows(c, h->d2, h->cond_cat, ldc, 0, h->txt_d2, 2 * n4, n4);
        HIPCHK(hipMemcpyAsync(h->x, x_T, (size_t)B * T * NF2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (h->cfg.single_only == 1) {
        const int cw = h->cond_w;                            // text_dim, or the latent size for MDMDenoiser (cond is added to the timestep embedding)
        HIPCHK(hipMemsetAsync(h->cond_cat, 0, (size_t)n * cw * sizeof(float), st));
        HIPCHK(hipMemcpyAsync(h->cond_cat, cond, (size_t)B * cw * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (h->d1.kind == 0) rc = linear(c, h->cond_cat, td, h->d1.te_w, td, h->d1.te_b, h->txt_d1, h->d1.st.D, n, h->d1.st.D, td);
        RC(load_x(h->x, NF));
    } else {
        // mixer: cond [B, 6*td + 2*td1] (mixermdm.py:342-354); dual: cond [B, 5*td] (in2in.py:299-305) -- same column order for the
        // first five slices, so text_all serves both
        const int cw = h->cond_w;
        HIPCHK(hipMemsetAsync(h->cond_cat, 0, (size_t)n * cw * sizeof(float), st));
        HIPCHK(hipMemcpyAsync(h->cond_cat, cond, (size_t)B * cw * sizeof(float), hipMemcpyDeviceToDevice, st));
        rc = text_all(c, h->cond_cat, n);
        RC(load_x(h->x, NF2));
        if (h->x2) RC(load_x(h->x2, NF2));                  // img2 = img.clone()
    }
    if (rc) return herr(h, rc);
    RC(mmdm_set_step(h->d_step, h->d_step + 1, h->S - 1, 0, st));
    h->host_step = h->S - 1;
    h->B = B; h->T = T; h->begun = true;
    h->geom = g;
    h->call_stream = st;
    // a new call keeps no history until mmdm_set_history says so: the descriptor is rewritten ON THE STREAM, behind any step of the
    // previous call still in flight, so an earlier call's buffers can never be written again (graphs do not bake them in)
    h->hist = mmdm_hist_desc{nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0};
    return herr(h, push_hist(h, st));
}

extern "C" int mmdm_begin(mmdm_handle h, const float* cond, const float* x_T, int B, int T, void* stream) {
    return begin_impl(h, cond, x_T, B, T, nullptr, stream);
}

extern "C" int mmdm_begin_ragged(mmdm_handle h, const float* cond, const float* x_T, int B, const int* lens_host, void* stream) {
    if (!lens_host) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_begin_ragged: null lengths");
    return begin_impl(h, cond, x_T, B, 0, lens_host, stream);
}

extern "C" int mmdm_call_rows(mmdm_handle h, int* rows, int* real_rows, int* ragged) {
    if (!h) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_call_rows: null handle");
    if (!h->begun) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_call_rows: call mmdm_begin / mmdm_begin_ragged first"));
    if (rows) *rows = h->geom.rows;
    if (real_rows) *real_rows = h->geom.real_rows;
    if (ragged) *ragged = h->geom.rag ? 1 : 0;
    return MMDM_OK;
}

extern "C" int mmdm_set_history(mmdm_handle h, float* influence_i1, float* influence_i2, float* out1, float* out2, float* out_influenced, int every) {
    if (!h) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_history: null handle");
    if (!h->begun) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_set_history: call after mmdm_begin"));
    if (every <= 0) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_set_history: every must be >= 1"));
    if (h->cfg.single_only != 0 && (influence_i1 || influence_i2 || out1 || out2 || out_influenced))
        return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_set_history: only the two-chain MixerMDM sampler has history side outputs"));
    h->hist = mmdm_hist_desc{influence_i1, influence_i2, out1, out2, out_influenced, every, 0};
    return herr(h, push_hist(h, h->call_stream));     // device-side descriptor: cached graphs are unaffected
}

extern "C" int mmdm_run(mmdm_handle h, int nsteps, int use_graph, void* stream) {
    if (!h) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_run: null handle");
    if (!h->begun) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_run: call mmdm_begin first"));
    if (nsteps < 0 || nsteps > h->host_step + 1)
        return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_run: %d steps requested, %d left in the schedule", nsteps, h->host_step + 1));
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ctx c{h, st, &h->sa, &h->geom};
    if (use_graph && !h->prof.on && !st) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_run: graph capture needs a non-default stream"));
    if (nsteps == 0) return MMDM_OK;
    // MMDM_SERIALIZE_HANDLES=1: this call's steps start behind the previous sampling call of any handle, and the next one behind them
    struct Serial {
        hipStream_t st; bool on;
        std::unique_lock<std::mutex> lk;
        explicit Serial(hipStream_t s) : st(s), on(g_serialize_handles.load() == 1) {
            if (!on) return;
            lk = std::unique_lock<std::mutex>(g_serial_mu);
            if (g_serial_ev) (void)hipStreamWaitEvent(st, g_serial_ev, 0);
        }
        ~Serial() {
            if (!on) return;
            if (!g_serial_ev) (void)hipEventCreateWithFlags(&g_serial_ev, hipEventDisableTiming);
            if (g_serial_ev) (void)hipEventRecord(g_serial_ev, st);
        }
    } serial(st);
    if (use_graph && !h->prof.on) {
        hipGraphExec_t exec = nullptr;
        hipEvent_t done = nullptr;
        int first = 0;
        // what a captured node bakes in: uniform (B, T, S); ragged (B, query tiles of the longest item, S, group stride) -- the lengths are device data
        const int kT = h->geom.rag ? (h->T + 63) / 64 : h->T, kR = h->geom.rag ? h->geom.rows : 0;
        for (auto& g : h->graphs)
            if (g.B == h->B && g.T == kT && g.S == h->S && g.rows == kR) { exec = g.exec; done = g.done; g.used = ++h->graph_clock; break; }
        if (!exec) {
            hipGraph_t g = nullptr;
            // one capture at a time in the process: several handles may be driven from several host threads (mmdm_create_shared), and two
            // concurrent captures take this runtime down; replays and eager launches of other handles go on beside a capture (relaxed mode)
            std::unique_lock<std::shared_mutex> capture_lock(g_graph_mu);
            HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            int rc = run_step(c);
            hipError_t e = hipStreamEndCapture(st, &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return herr(h, rc); }
            if (e != hipSuccess) return herr(h, mmdm_set_error(MMDM_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e)));
            e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) return herr(h, mmdm_set_error(MMDM_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)));
            if (h->graphs.size() >= h->graph_cap && !must_park(h)) {      // evict the least recently used entry (never while execs are parked rather than destroyed: retire_exec)
                size_t lru = 0;
                for (size_t i = 1; i < h->graphs.size(); ++i)
                    if (h->graphs[i].used < h->graphs[lru].used) lru = i;
                // its replays may still be queued, possibly on another stream than this call's: wait for the event recorded behind its LAST
                // replay -- not for the whole device, which would stall other handles' streams
                e = hipEventSynchronize(h->graphs[lru].done);
                if (e != hipSuccess) {
                    (void)hipGraphExecDestroy(exec);
                    return herr(h, mmdm_set_error(MMDM_ERR_HIP, "mmdm_run: waiting for the last replay of the graph being evicted: %s", hipGetErrorString(e)));
                }
                retire_exec(h, h->graphs[lru].exec);
                (void)hipEventDestroy(h->graphs[lru].done);
                h->graphs.erase(h->graphs.begin() + lru);
            }
            e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
            if (e != hipSuccess) {
                (void)hipGraphExecDestroy(exec);
                return herr(h, mmdm_set_error(MMDM_ERR_HIP, "mmdm_run: hipEventCreate: %s", hipGetErrorString(e)));
            }
            h->graphs.push_back({h->B, kT, h->S, kR, exec, ++h->graph_clock, done});
            ++h->n_captures;
            // the FIRST launch of a fresh exec binds the runtime's internal branch streams to it (hip::Graph::UpdateStreams): still inside the
            // exclusive section -- beside another thread's launch that is where the runtime was seen to crash
            HIPCHK(hipGraphLaunch(exec, st));
            first = 1;
        }
        {
            std::shared_lock<std::shared_mutex> launch_lock(g_graph_mu);
            for (int k = first; k < nsteps; ++k) HIPCHK(hipGraphLaunch(exec, st));
            HIPCHK(hipEventRecord(done, st));
        }
        h->n_replays += nsteps;
    } else {
        for (int k = 0; k < nsteps; ++k) {
            int rc = run_step(c);
            if (rc) return herr(h, rc);
        }
    }
    h->host_step -= nsteps;
    return MMDM_OK;
}

extern "C" int mmdm_seek(mmdm_handle h, int step_index, void* stream) {
    if (!h) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_seek: null handle");
    if (!h->begun) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_seek: call mmdm_begin first"));
    if (step_index < 0 || step_index >= h->S) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_seek: step index %d outside [0, %d)", step_index, h->S));
    RC(mmdm_set_step(h->d_step, h->d_step + 1, step_index, h->S - 1 - step_index, static_cast<hipStream_t>(stream)));
    h->host_step = step_index;
    return MMDM_OK;
}

extern "C" int mmdm_get_state(mmdm_handle h, float** x, float** x2, float** pred_xstart, float** pred_xstart2, float** model_out) {
    if (!h) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_get_state: null handle");
    if (x) *x = h->x;
    if (x2) *x2 = h->x2;
    if (pred_xstart) *pred_xstart = h->px1;
    if (pred_xstart2) *pred_xstart2 = h->px2;
    if (model_out) *model_out = h->model_out;
    return MMDM_OK;
}

extern "C" int mmdm_copy_result(mmdm_handle h, float* dst, void* stream) {
    if (!h || !dst) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_copy_result: null argument");
    if (!h->begun) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_copy_result: call mmdm_begin / mmdm_begin_ragged first"));
    const float* src = h->cfg.single_only ? h->px1 : h->px2;       // the loop's return value: last pred_xstart2 (two-chain) / pred_xstart (single chain)
    const size_t n = (size_t)h->geom.real_rows * (h->cfg.single_only == 1 ? NF : NF2);
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return MMDM_OK;
}

extern "C" int mmdm_module_forward(mmdm_handle h, int which, const float* x, const float* x2, const float* cond, int t,
                                   float* out, int n, int T, void* stream) {
    if (!h || !x || !cond || !out) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_module_forward: null argument");
    if (!h->prepared) return herr(h, mmdm_set_error(MMDM_ERR_STATE, "mmdm_module_forward: call mmdm_prepare first"));
    const int so_ = h->cfg.single_only;
    const bool cfgx2 = which == 4;               // ClassifierFreeSampleModelX2.forward: n = B rows in, B rows out
    if (cfgx2) {
        if (so_ != 0) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_module_forward: module 4 (CFG x2) needs the two-chain handle"));
        if (n <= 0 || n > h->cfg.max_batch || T <= 0 || T > h->cfg.max_frames)
            return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_module_forward: B=%d (<= max_batch) T=%d out of range", n, T));
    } else if (n <= 0 || (n & 1) || n / 2 > h->cfg.max_batch * (so_ == 2 ? 2 : 1) || T <= 0 || T > h->cfg.max_frames)
        return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_module_forward: n=%d (even, <= 2*max_batch) T=%d out of range", n, T));
    if (t < 0 || t >= 5000) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_module_forward: timestep %d out of range", t));
    if (which < 0 || which > 4 || (so_ == 1 && which != 0) || (so_ == 2 && which != 1) || (so_ == 3 && (which == 2 || which == 0)) || (so_ != 3 && which == 3)) return herr(h, mmdm_set_error(MMDM_ERR_ARG, "mmdm_module_forward: bad module %d", which));
    hipStream_t st = static_cast<hipStream_t>(stream);
    Ctx c{h, st, &h->sa};
    const int td = h->cfg.text_dim;
    ProfPause pause(h->prof);
    TailScope tail_scope((so_ == 1 || so_ == 2) ? 10 : 0);
    // The forward borrows slot 0 of the schedule tables (timestep_map[0] and row 0 of every time_tab) for a one-entry schedule
    // time_tab[0] = time_embed(pe[t]) and puts the caller's entries back afterwards, so a schedule set before survives the call
    // (a sampling call in progress does not: text embeddings, scratch and the step index are overwritten -- mmdm_begin again).
    const int S_keep = h->S;
    ModuleW* mods[3] = {so_ != 2 ? &h->d1 : nullptr, so_ != 1 ? &h->d2 : nullptr, so_ == 0 ? &h->mx : nullptr};
    const size_t Dst = max2(max2(h->d1.st.D, h->d2.st.D), h->mx.st.D);
    if (S_keep > 0)
        for (int k = 0; k < 3; ++k)
            if (mods[k]) HIPCHK(hipMemcpyAsync(h->tt_stash + k * Dst, mods[k]->time_tab, mods[k]->st.D * sizeof(float), hipMemcpyDeviceToDevice, st));
    int rc = MMDM_OK;
    // every exit from here on goes through `done`, which puts S, timestep_map[0] and row 0 of the time tables back
    auto done = [&](int code) {
        h->S = S_keep;
        if (S_keep > 0) {
            hipError_t e = hipMemcpyAsync(h->d_tmap, &h->h_tmap0, sizeof(int), hipMemcpyHostToDevice, st);
            for (int k = 0; k < 3 && e == hipSuccess; ++k)
                if (mods[k]) e = hipMemcpyAsync(mods[k]->time_tab, h->tt_stash + k * Dst, mods[k]->st.D * sizeof(float), hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess && !code) code = mmdm_set_error(MMDM_ERR_HIP, "mmdm_module_forward: restoring the schedule tables failed: %s", hipGetErrorString(e));
        }
        return herr(h, code);
    };
    h->S = 1;
    h->begun = false;
    {
        hipError_t e = hipMemcpyAsync(h->d_tmap, &t, sizeof(int), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return done(mmdm_set_error(MMDM_ERR_HIP, "mmdm_module_forward: timestep upload failed: %s", hipGetErrorString(e)));
        if ((rc = mmdm_set_step(h->d_step, h->d_step + 1, 0, 0, st))) return done(rc);
    }
    if (which == 3) {                         // in2INDenoiser "dual_individual": x [n,T,524], cond [n, 5*td]
        if ((rc = build_time_tab(c, h->d1))) return done(rc);
        if ((rc = text_rows(c, h->d1, cond, 5 * td, 3 * td, h->txt_d1, 0, n))) return done(rc);
        if ((rc = text_rows(c, h->d1, cond, 5 * td, 4 * td, h->txt_d1, n, n))) return done(rc);
        if ((rc = cond_vectors(c, h->d1, h->txt_d1, h->se_d1, h->ss_d1, 2 * n))) return done(rc);
        rc = run_dual_individual(c, h->d1, x, n, n, T, h->ss_d1, ss_ld_of(h->d1), out);
        return done(rc);
    }
    if (which == 0 && h->d1.kind == 1) {      // MDMDenoiser.forward: cond [n, latent]
        if ((rc = build_time_tab(c, h->d1))) return done(rc);
        rc = run_denoiser_mdm(c, h->d1, x, n, 1, NF, n, T, cond, h->td1, out, NF);
        return done(rc);
    }
    if (which == 0) {
        if ((rc = build_time_tab(c, h->d1))) return done(rc);
        if ((rc = linear(c, cond, td, h->d1.te_w, td, h->d1.te_b, h->txt_d1, h->d1.st.D, n, h->d1.st.D, td))) return done(rc);
        if ((rc = cond_vectors(c, h->d1, h->txt_d1, h->se_d1, h->ss_d1, n))) return done(rc);
        rc = run_denoiser(c, h->d1, false, x, n, 1, NF, n, T, h->ss_d1, ss_ld_of(h->d1), out, NF);
        return done(rc);
    }
    if (which == 1) {
        if ((rc = build_time_tab(c, h->d2))) return done(rc);
        const bool ig = h->cfg.model2_kind == 1;
        const int ldc = 3 * td;
        if ((rc = text_rows(c, h->d2, cond, ldc, ig ? 0 : td, h->txt_d2, 0, n))) return done(rc);
        if ((rc = text_rows(c, h->d2, cond, ldc, ig ? 0 : 2 * td, h->txt_d2, n, n))) return done(rc);
        if ((rc = text_rows(c, h->d2, cond, ldc, 0, h->txt_d2, 2 * n, n))) return done(rc);
        if ((rc = cond_vectors(c, h->d2, h->txt_d2, h->se_d2, h->ss_d2, 3 * n))) return done(rc);
        rc = run_denoiser(c, h->d2, true, x, n, 2, NF2, n, T, h->ss_d2, ss_ld_of(h->d2), out, NF2);
        return done(rc);
    }
    if (!x2) return done(mmdm_set_error(MMDM_ERR_ARG, "mmdm_module_forward: Mixer.forward needs x2"));
    if ((rc = build_time_tab(c, h->d1)) || (rc = build_time_tab(c, h->d2)) || (rc = build_time_tab(c, h->mx))) return done(rc);
    // which == 2: Mixer.forward on a batch the caller has already CFG-doubled (n rows of x / x2 / cond).
    // which == 4: ClassifierFreeSampleModelX2.forward (cfg_sampler.py:38-56): B = n rows in; the doubling (x repeated, cond followed by
    //             zero rows) is this library's row layout, the s*cond + (1-s)*uncond combine is fused in the blend kernel.
    const int nn = cfgx2 ? 2 * n : n, xb = n;
    const float* cnd = cond;
    if (cfgx2) {
        const int cw = h->cond_w;
        hipError_t e = hipMemsetAsync(h->cond_cat, 0, (size_t)nn * cw * sizeof(float), st);
        if (e == hipSuccess) e = hipMemcpyAsync(h->cond_cat, cond, (size_t)n * cw * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return done(mmdm_set_error(MMDM_ERR_HIP, "mmdm_module_forward: %s", hipGetErrorString(e)));
        cnd = h->cond_cat;
    }
    if ((rc = text_all(c, cnd, nn))) return done(rc);
    if (h->d1.kind == 0 && (rc = cond_vectors(c, h->d1, h->txt_d1, h->se_d1, h->ss_d1, 2 * nn))) return done(rc);
    if ((rc = cond_vectors(c, h->d2, h->txt_d2, h->se_d2, h->ss_d2, 3 * nn))) return done(rc);
    if ((rc = cond_vectors(c, h->mx, h->txt_mx, h->se_mx, h->ss_mx, 3 * nn))) return done(rc);
    rc = h->d1.kind == 1 ? run_denoiser_mdm(c, h->d1, x, xb, 2, NF2, nn, T, cnd + 3 * td, h->cond_w, h->o1, NF2)
                         : run_denoiser(c, h->d1, false, x, xb, 2, NF2, nn, T, h->ss_d1, ss_ld_of(h->d1), h->o1, NF2);
    if (rc) return done(rc);
    if ((rc = run_denoiser(c, h->d2, true, x2, xb, 2, NF2, nn, T, h->ss_d2, ss_ld_of(h->d2), h->o2, NF2))) return done(rc);
    // Mixer.forward returns out_influenced for the whole CFG-doubled batch: it is the blend kernel's out_influenced history output
    // (slot 0 of a one-step history); the CFG wrapper returns the combined rows the same kernel leaves in model_out.
    h->hist = mmdm_hist_desc{nullptr, nullptr, nullptr, nullptr, cfgx2 ? nullptr : out, 1, 0};
    if ((rc = push_hist(h, st))) return done(rc);
    rc = mixer_core(c, nn / 2, T, true);
    h->hist.mix = nullptr;
    if (!rc) rc = push_hist(h, st);
    if (!rc && cfgx2) {
        hipError_t e = hipMemcpyAsync(out, h->model_out, (size_t)n * T * NF2 * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) rc = mmdm_set_error(MMDM_ERR_HIP, "mmdm_module_forward: %s", hipGetErrorString(e));
    }
    return done(rc);
}

extern "C" int mmdm_graph_parked(void) {
    std::shared_lock<std::shared_mutex> lock(g_graph_mu);
    return (int)g_parked_execs.size();
}

extern "C" int mmdm_graph_stats(mmdm_handle h, int64_t* captures, int64_t* replays, int* cached) {
    if (!h) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_graph_stats: null handle");
    if (captures) *captures = h->n_captures;
    if (replays) *replays = h->n_replays;
    if (cached) *cached = (int)h->graphs.size();
    return MMDM_OK;
}

extern "C" size_t mmdm_encoder_layer_workspace(int nseq, int T, int D, int F) {
    if (nseq <= 0 || T <= 0 || D <= 0 || F <= 0) return 0;
    return (size_t)nseq * T * ((size_t)5 * D + F);
}

extern "C" int mmdm_encoder_layer_f32(float* x, const mmdm_encoder_layer_weights* w, int nseq, int T, int D, int H, int F, int norm_first,
                                      int activation, int causal, float eps, float* workspace, size_t workspace_floats, void* stream) {
    if (nseq == 0 || T == 0) return MMDM_OK;
    if (!x || !w || !workspace || nseq < 0 || T < 0 || D <= 0 || H <= 0 || D % H || F <= 0)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_encoder_layer_f32: bad arguments nseq=%d T=%d D=%d H=%d F=%d", nseq, T, D, H, F);
    if (activation != MMDM_EPI_BIAS_GELU && activation != MMDM_EPI_BIAS_QUICKGELU)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_encoder_layer_f32: activation must be MMDM_EPI_BIAS_GELU or MMDM_EPI_BIAS_QUICKGELU");
    if (workspace_floats < mmdm_encoder_layer_workspace(nseq, T, D, F))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_encoder_layer_f32: workspace too small (%zu < %zu floats)", workspace_floats, mmdm_encoder_layer_workspace(nseq, T, D, F));
    for (const float* q : {w->in_proj_weight, w->in_proj_bias, w->out_proj_weight, w->out_proj_bias, w->linear1_weight, w->linear1_bias,
                           w->linear2_weight, w->linear2_bias, w->norm1_weight, w->norm1_bias, w->norm2_weight, w->norm2_bias})
        if (!q) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_encoder_layer_f32: null weight pointer");
    RC(mmdm_kernels_init());
    const size_t R = (size_t)nseq * T;
    float* qkv = workspace;
    float* att = qkv + R * 3 * D;
    float* tmp = att + R * D;
    float* f1 = tmp + R * D;
    EncLayerW e;
    auto nc = [](const float* q) { return const_cast<float*>(q); };
    e.in_w = nc(w->in_proj_weight); e.in_b = nc(w->in_proj_bias); e.out_w = nc(w->out_proj_weight); e.out_b = nc(w->out_proj_bias);
    e.l1_w = nc(w->linear1_weight); e.l1_b = nc(w->linear1_bias); e.l2_w = nc(w->linear2_weight); e.l2_b = nc(w->linear2_bias);
    e.n1_g = nc(w->norm1_weight); e.n1_b = nc(w->norm1_bias); e.n2_g = nc(w->norm2_weight); e.n2_b = nc(w->norm2_bias);
    Ctx c{nullptr, static_cast<hipStream_t>(stream), nullptr};
    return encoder_layer(c, x, e, nseq, T, D, H, F, norm_first != 0, activation, causal != 0, eps, qkv, att, tmp, f1);
}

extern "C" int mmdm_profile_enable(mmdm_handle h, int on) {
    if (!h) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_profile_enable: null handle");
    h->prof.on = on != 0;
    if (on)
        for (int k = 0; k < Prof::NCLS; ++k) { h->prof.used[k] = 0; h->prof.flops[k] = 0; h->prof.bytes[k] = 0; }
    return MMDM_OK;
}

extern "C" int mmdm_profile_read(mmdm_handle h, int which, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    if (!h || which < 0 || which >= Prof::NCLS) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_profile_read: bad argument");
    Prof& p = h->prof;
    HIPCHK(hipDeviceSynchronize());
    double ms = 0;
    for (size_t i = 0; i + 1 < p.used[which]; i += 2) {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, p.ev[which][i], p.ev[which][i + 1]));
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (int64_t)(p.used[which] / 2);
    if (flops) *flops = p.flops[which];
    if (bytes) *bytes = p.bytes[which];
    return MMDM_OK;
}
