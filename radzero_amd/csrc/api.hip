// radzero_hip — C-ABI (include/radzero_hip.h): handle, checkpoint packing, forward orchestration.
#include "../../include/radzero_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rz_kernels.h"

using namespace rz;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
// Tuning / A-B switches.  Every handle owns a set (rz_set_model_option); a field left at RZ_OPT_INHERIT follows the process-wide value
// (rz_set_option: what the measurement tools flip), so two handles of one process can differ and a test that sets a handle's option
// cannot leak into another handle.
constexpr int RZ_OPT_INHERIT = INT32_MIN;
struct Options {
    int vision_chunk;     // images per pass of rz_vision_forward (0 = whole batch)
    int vision_streams;   // 2 = split the batch over two internal HIP streams
    int mlp_chunk;        // images per fc1->fc2 pass (0 = whole batch = default, -1 = auto ~126 MiB of hidden rows): run_chunk in rz_vision_forward
    int attn_variant;     // 0 = default; see include/radzero_hip.h
    int gemm_variant;     // 0 = auto; see include/radzero_hip.h
    int gemm_f32_split;   // fp32 mode: the vision encoder's GEMMs on the f16 matrix pipe with hi/lo-split operands (0 = exact-fp32 MFMAs)
    int attn_f32_split;   // fp32 mode: attention on the f16 matrix pipe with hi/lo-split operands (0 = exact-fp32 MFMAs)
    int ln_fused;         // 1 = fuse the blocks' LayerNorms into the GEMMs either side where the persistent kernel applies (16-bit modes)
    int sim_op;           // VL-CABS similarity: 0 = "cos" (released config), 1 = "dot" (losses.py:214-215)
    int pad_rows;         // token rows per image: 0 = multiple of 128, of 256 when that costs < 2 % more rows | 128 | 256 = always that multiple
    int f32_split_guard;  // fp32 mode: 1 = a forward whose f16 planes overflowed is repeated on the exact-fp32 kernels (default)
    int gemm_f32_mx;      // fp32 mode: 1 (default) = the split GEMMs' two correction terms run as one block-scaled fp8 MFMA (MX form, rz_common.h) wherever the launch's rows are a multiple of 256; 2 = the same; 3 = only from 64 row tiles of 256 on (the default of rounds 4-5); 0 = three f16 planes
    int attn_f32_mx;      // fp32 mode, with the MX GEMM form: 1 (default) = the attention's P V correction terms as block-scaled fp8 MFMAs, scores at 22 bits; 2 = scores too; 0 = f16 planes
    int gemm_raster;      // gemm12.hip: tile order inside an XCD (GemmArgs::raster): 0 = 4 x tiles_n groups | S > 0 = slab walk, <= S n tiles per slab
    int attn_f32_pv;      // fp32 mode: 1 = the attention's P V product on the hi planes alone (f16 P and V, row sums of the rounded P on the matrix pipe); 0 = with its correction terms ("f32_precision high")
    int gemm_small_tile;  // 128x128-kernel family's tile (GemmArgs::small_tile): 0 = by grid size | 1 128 x 128 | 2 64 x 64 | 3 128 x 64; + 20 / + 40 = two / four LDS stages
    int gemm_qkv_pair;    // 16-bit modes, small batches: 1 (default) = the q|k and v projections of a block as ONE launch of the 128x128 kernel family (gemm_pair_kernel) where the persistent kernel's merged projection does not apply; 0 = two launches
    int f32_drop;         // fp32 mode, ACCURACY ABLATION (tools/fp32_term_ablation.py; three-plane form): bit mask of product classes computed hi . hi only — 1 q|k projection, 2 V projection, 4 out-projection, 8 fc1, 16 fc2, 32 patch embedding; tools build also: 64 scores, 128 P (V keeps hi + lo)
};
Options g_opt = {0, 1, 0, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 0, 1, 0};
const Options kInherit = {RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT,
                          RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT,
                          RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT, RZ_OPT_INHERIT};
int* option_field(Options& o, const char* name) {
#ifdef RZ_EXPERIMENTS      // measured, never a gain (profiles/NOTEBOOK.md): known to the tools build only; the product runs one pass, one stream
    if (!strcmp(name, "vision_chunk")) return &o.vision_chunk;
    if (!strcmp(name, "vision_streams")) return &o.vision_streams;
    if (!strcmp(name, "mlp_chunk")) return &o.mlp_chunk;
    if (!strcmp(name, "gemm_raster")) return &o.gemm_raster;
#endif
    if (!strcmp(name, "attn_variant")) return &o.attn_variant;
    if (!strcmp(name, "gemm_variant")) return &o.gemm_variant;
    if (!strcmp(name, "gemm_small_tile")) return &o.gemm_small_tile;
    if (!strcmp(name, "gemm_qkv_pair")) return &o.gemm_qkv_pair;
    if (!strcmp(name, "gemm_f32_split")) return &o.gemm_f32_split;
    if (!strcmp(name, "attn_f32_split")) return &o.attn_f32_split;
    if (!strcmp(name, "ln_fused")) return &o.ln_fused;
    if (!strcmp(name, "sim_op")) return &o.sim_op;
    if (!strcmp(name, "pad_rows")) return &o.pad_rows;
    if (!strcmp(name, "f32_split_guard")) return &o.f32_split_guard;
    if (!strcmp(name, "gemm_f32_mx")) return &o.gemm_f32_mx;
    if (!strcmp(name, "attn_f32_mx")) return &o.attn_f32_mx;
    if (!strcmp(name, "attn_f32_pv")) return &o.attn_f32_pv;
    if (!strcmp(name, "f32_drop")) return &o.f32_drop;
    return nullptr;
}
inline int pick(int own, int process_wide) { return own == RZ_OPT_INHERIT ? process_wide : own; }
// gemm_small_tile: geometry (0 rule | 1 | 2 | 3) + 10 x stages (0 rule | 2 | 4)
inline bool small_tile_ok(int value) {
    if (value == RZ_OPT_INHERIT) return true;
    if (value < 0) return false;
    const int geo = value % 10, st = value / 10;
    return geo <= 3 && (st == 0 || st == 2 || st == 4);
}
inline bool f32_drop_ok(int value) {
#ifdef RZ_EXPERIMENTS
    (void)value;
    return true;
#else
    return value == RZ_OPT_INHERIT || (value & (64 | 128)) == 0;
#endif
}

hipError_t flash_attn(int variant, int dt, const void* q, const void* k, const void* vt, void* ctx, int64_t bs, int B, int H, int nv, int np,
                      hipStream_t s, const unsigned* run_if = nullptr) {
    return launch_flash_attn(dt, q, k, vt, ctx, bs, B, H, nv, np, variant, s, run_if);       // unknown values run the default shapes
}

int hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return (int)e;
}
#define RZ_HIP(x)                                          \
    do {                                                   \
        hipError_t _e = (x);                               \
        if (_e != hipSuccess) return hip_fail(_e, #x);     \
    } while (0)

size_t dsize(int dt) { return dt == RZ_F32 ? 4 : 2; }

// fp32 -> compute dtype on the host (round-to-nearest-even; NaN-safe enough for weights)
uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
uint16_t f32_to_f16(float f) {
    _Float16 h = (_Float16)f;
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }            // a buffer added to rz_model can no longer be forgotten in rz_destroy
    hipError_t ensure(size_t n, bool zero) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) return e;
        bytes = n;
        if (zero) e = hipMemset(p, 0, n);
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct Tensor {          // one packed checkpoint tensor on the device
    void* p = nullptr;
    void* p3 = nullptr;  // fp32 mode, matrices: [N][3K] f16 planes [hi | hi | lo] for the hi/lo-split GEMMs (rz_weights_ready)
    void* p4 = nullptr;  // fp32 mode, matrices: the MX form [N][K f16 | per 64 columns: hi8 x 64, lo8 x 64] (4 K bytes per row; rz_common.h)
    bool loaded = false;
};

struct DinoBlock {       // TF:dinov2/modeling_dinov2.py:342-380
    Tensor ln1_g, ln1_b, wqkv, bqkv, wo, bo, ls1, ln2_g, ln2_b, w1, b1, w2, b2, ls2;   // wqkv: [3D][D] rows q | k | v
    int qkv_parts = 0;   // bit mask of loaded q/k/v weight (1,2,4) and bias (8,16,32) pieces
    // fused LayerNorm (gemm8.hip): c1[n] = sum_k gamma[k] W[n][k] over the ROUNDED weights, c2 = W beta + b; built from the host
    // copies below (16-bit modes only), which STAY for the life of the handle (16.5 MB of host memory per block, ~200 MB for 12 + 2
    // blocks) so that a partial reload — one rz_load_weight of a gain, say — re-folds from complete fp32 data (load_block)
    Tensor c1qkv, c2qkv, c1_1, c2_1;
    std::vector<float> h_wqkv, h_bqkv, h_w1, h_b1, h_g1, h_be1, h_g2, h_be2;
    bool folded = false;
};
struct TextLayer {       // TF:mpnet/modeling_mpnet.py:234-261
    Tensor wqkv, bqkv, wo, bo, lna_g, lna_b, w1, b1, w2, b2, lno_g, lno_b;
    int qkv_parts = 0;
};

}  // namespace

struct rz_model {
    rz_config cfg;
    Options opt = kInherit;
#ifdef RZ_EXPERIMENTS
    int o_vision_chunk() const { return pick(opt.vision_chunk, g_opt.vision_chunk); }
    int o_vision_streams() const { return pick(opt.vision_streams, g_opt.vision_streams); }
    int o_mlp_chunk() const { return pick(opt.mlp_chunk, g_opt.mlp_chunk); }
#else
    int o_vision_chunk() const { return 0; }
    int o_vision_streams() const { return 1; }
    int o_mlp_chunk() const { return 0; }
#endif
    int o_attn_variant() const { return pick(opt.attn_variant, g_opt.attn_variant); }
    int o_gemm_variant() const { return pick(opt.gemm_variant, g_opt.gemm_variant); }
    bool o_gemm_qkv_pair() const { return pick(opt.gemm_qkv_pair, g_opt.gemm_qkv_pair) != 0; }
    // the pair in front of the persistent kernel's merged projection: measured in the step (profiles/r06/qkv_pair_step_ab2.txt) the pair wins up to 432 tiles of 128 x 128
    // (224^2 x 4 / x 8 -7 / -3 %, 518^2 x 1 / x 2 -6 / -3 %) and loses from 648 on (518^2 x 3 +6 %, 1024^2 x 1 +5 %); option value 2 = wherever the pair applies (A/B)
    bool o_gemm_qkv_pair_first(int M) const {
        const int v = pick(opt.gemm_qkv_pair, g_opt.gemm_qkv_pair);
        return v == 2 || (v == 1 && o_gemm_variant() == 0 && (int64_t)(M / 128) * (3 * D / 128) <= 448);
    }
    int o_gemm_small_tile() const { return pick(opt.gemm_small_tile, g_opt.gemm_small_tile); }
    int o_gemm_raster() const { return pick(opt.gemm_raster, g_opt.gemm_raster); }
    int o_gemm_f32_mx() const { return pick(opt.gemm_f32_mx, g_opt.gemm_f32_mx); }
    int o_attn_f32_mx() const { return pick(opt.attn_f32_mx, g_opt.attn_f32_mx); }
    int o_attn_f32_pv() const { return pick(opt.attn_f32_pv, g_opt.attn_f32_pv); }
    int o_f32_drop() const { return pick(opt.f32_drop, g_opt.f32_drop); }
    bool o_gemm_f32_split() const { return pick(opt.gemm_f32_split, g_opt.gemm_f32_split) != 0 && !force_exact; }
    bool o_attn_f32_split() const { return pick(opt.attn_f32_split, g_opt.attn_f32_split) != 0 && !force_exact; }
    bool o_ln_fused() const { return pick(opt.ln_fused, g_opt.ln_fused) != 0; }
    int o_pad_rows() const { return pick(opt.pad_rows, g_opt.pad_rows); }
    bool o_sim_dot() const { return pick(opt.sim_op, g_opt.sim_op) == 1; }
    bool o_guard() const { return pick(opt.f32_split_guard, g_opt.f32_split_guard) != 0; }
    bool force_exact = false;            // set while the exact-fp32 pass of a guarded forward is enqueued (overflow guard)
    const unsigned* run_if = nullptr;    // that pass's predicate (the guard word): its kernels do nothing unless the word is set
    // padded token rows per image: a multiple of 128 (every GEMM M-tile and attention query block is full); of 256 when that costs less
    // than 2 % more rows, so that the 256x256 GEMM kernels apply at any batch size (one 1536^2 image: 11882 -> 12032 instead of 11904).
    // Round 5: also of 256 for an ODD batch whose 128-multiple is an odd one (B x rows would not be a multiple of 256: every GEMM on the 128 x 128
    // kernels) when that costs <= 10 % more rows — 518^2: 1408 -> 1536 for 1, 3, 5, ... images (+6 % / +13 % / +9 % / +12 % images per second at
    // 1 / 3 / 5 / 7 images, profiles/NOTEBOOK.md r5).  Pad rows are masked keys and unused query rows: the results' bits do not depend on the choice
    // (tests: forced 256-row padding, batches 1..10).  batch = 0: the batch-independent upper bound (table and workspace sizes).
    int pad_tokens(int nv, int batch = 0) const {
        const int p128 = (nv + 127) / 128 * 128, p256 = (nv + 255) / 256 * 256, rule = o_pad_rows();
        if (rule == 128) return p128;
        if (rule == 256) return p256;
        if ((p256 - nv) * 50 <= nv) return p256;
        const bool odd_case = p128 != p256 && (p256 - p128) * 10 <= p128;
        if (odd_case && (batch == 0 || ((batch & 1) && (int64_t)batch * p256 >= 4 * 256))) return p256;
        return p128;
    }
    int dt;               // compute dtype
    int D, H, F, KP, KPAD;
    std::vector<DinoBlock> blocks;
    std::vector<TextLayer> tlayers;
    Tensor patch_w, vit_ln_g, vit_ln_b, word_emb, pos_emb, temb_ln_g, temb_ln_b, shared_ln_g, shared_ln_b;
    std::vector<float> cls_host, patch_bias_host;
    bool cls_loaded = false, patch_bias_loaded = false, tau_loaded = false, attn_tau_loaded = false;
    bool tau_session = false, attn_tau_session = false;      // loaded since the last rz_weights_ready (a checkpoint WITHOUT attn_temperature resets it)
    float tau = 0.07f, attn_tau = 0.07f;      // exp(loss_temperature); exp(attn_temperature) when the checkpoint carries one (losses.py:57-63)
    // position tables per grid: [n_pad][D] fp32 = pos (+cls | +conv bias), zero on pad rows
    struct PosTable { DevBuf buf; int n_valid, n_pad; };
    std::map<std::pair<int, int>, PosTable> pos_tables;
    // workspaces
    int cap_batch = 0, cap_npad = 0, cap_trows = 0, cap_prompts = 0;
    DevBuf h, xn, qk, vt, ctx, mid, vws, qhat, lnpart, lnstat, lnmu;      // xn doubles as the residual's T copy on the fused-LayerNorm path
    DevBuf th, txn, tqkv, tctx, tmid, tsum;
    DevBuf asplit;                       // fp32 mode: [hi | lo | hi] f16 planes of the A operand of the GEMM in flight (3 x max K per token row)
    DevBuf tasplit;                      // the same for the text encoder's GEMMs (round 6: its fp32 GEMMs run on the three-plane f16 form too)
    bool text_split_ok = false;          // the text encoder's matrices have [hi | hi | lo] copies
    DevBuf ovf;                          // fp32 mode, 8 words: [0] the hi/lo-split producers OR into it when a value leaves a plane's range, [1..3] weight checks
                                         // (rz_weights_ready), [4] forwards repeated on the exact-fp32 kernels since rz_create (counted on the device), [5] / [6] the same flag and
                                         // counter for the TEXT encoder (its own words: a prompt encode on a side stream must not trip the vision forward's guard)
    unsigned* ovf_host = nullptr;        // pinned mirror of words [0..3] (weight checks only: no forward reads it)
    struct SplitW { const char* p; size_t bytes; const char* p3; const char* p4; int e8_hi; };      // e8_hi: E8M0 scale byte of the MX copy's hi8 plane (lo8: 11 below)
    std::vector<SplitW> split_w;         // fp32 weight matrix -> its split copy
    bool split_dirty = true;             // a weight was (re)loaded since the split copies were built
    bool mx_weights_ok = false;          // every split weight fits the MX form's hi8 plane
    // state of the last vision forward
    int last_batch = 0, last_nvalid = 0, last_npad = 0;
    int last_f32_form = 0;               // operand form of the last forward's GEMMs: 0 the dtype's own kernels (16-bit modes, exact fp32), 1 three f16 planes, 2 MX form
    // profiling
    bool prof = false;
    unsigned prof_mask = 0x1F;   // kernel families that record events (bit = rz_prof_family)
    struct Ev { hipEvent_t a, b; int fam; };
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
    float prof_ms[RZ_PROF_NFAM] = {};
    int64_t prof_n[RZ_PROF_NFAM] = {};
    std::vector<void*> allocs;
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t side_done[2] = {nullptr, nullptr}, fork = nullptr;
    bool side_ready = false;

    hipError_t upload(Tensor& t, const float* src, size_t n, bool as_compute) {
        size_t es = as_compute ? dsize(dt) : 4;
        if (!t.p) {
            hipError_t e = hipMalloc(&t.p, n * es);
            if (e != hipSuccess) return e;
            allocs.push_back(t.p);
        }
        return upload_at(t.p, src, n, as_compute);
    }
    hipError_t upload_at(void* dst, const float* src, size_t n, bool as_compute) {
        if (!as_compute || dt == RZ_F32) return hipMemcpy(dst, src, n * 4, hipMemcpyHostToDevice);
        std::vector<uint16_t> tmp(n);
        if (dt == RZ_BF16) for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_bf16(src[i]);
        else for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_f16(src[i]);
        return hipMemcpy(dst, tmp.data(), n * 2, hipMemcpyHostToDevice);
    }
};

namespace {

struct ProfScope {
    rz_model* m;
    hipStream_t s;
    int idx = -1;
    ProfScope(rz_model* m_, int fam, hipStream_t s_) : m(m_), s(s_) {
        if (!m->prof || !((m->prof_mask >> fam) & 1u) || m->ev_used >= (1u << 20)) return;      // bounded: profiling left on without reads stops recording
        if (m->run_if) return;       // the overflow guard's predicated pass: ~75 launches that normally do nothing must not halve the families' average launch time
        if (m->ev_used == m->ev_pool.size()) {
            rz_model::Ev e;
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
            m->ev_pool.push_back(e);
        }
        idx = (int)m->ev_used++;
        m->ev_pool[idx].fam = fam;
        (void)hipEventRecord(m->ev_pool[idx].a, s);
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(m->ev_pool[idx].b, s);
    }
};

int round_up(int x, int m) { return (x + m - 1) / m * m; }

bool parse_layer(const char* name, const char* prefix, int* idx, const char** rest) {
    size_t n = strlen(prefix);
    if (strncmp(name, prefix, n) != 0) return false;
    char* end = nullptr;
    long v = strtol(name + n, &end, 10);
    if (end == name + n || *end != '.') return false;
    *idx = (int)v;
    *rest = end + 1;
    return true;
}

// copy `rows x cols` fp32 host block scaled by `scale` into a compute-dtype device matrix at row offset
int put_rows(rz_model* m, Tensor& t, size_t total_rows, size_t cols, size_t row_off, const float* src, size_t rows,
             float scale, bool as_compute) {
    size_t es = as_compute ? dsize(m->dt) : 4;
    if (!t.p) {
        RZ_HIP(hipMalloc(&t.p, total_rows * cols * es));
        m->allocs.push_back(t.p);
    }
    std::vector<float> tmp;
    const float* s = src;
    if (scale != 1.0f) {
        tmp.assign(src, src + rows * cols);
        for (auto& x : tmp) x *= scale;
        s = tmp.data();
    }
    RZ_HIP(m->upload_at((char*)t.p + row_off * cols * es, s, rows * cols, as_compute));
    return 0;
}

int load_dino_block(rz_model* m, DinoBlock& b, const char* rest, const float* data, int64_t numel) {
    const size_t D = m->D, F = m->F;
    // softmax scaling 1/sqrt(dh) AND log2(e) folded into q: the flash kernel works in log2 units (2^x softmax)
    const float qscale = 1.4426950408889634f / sqrtf((float)(m->D / m->H));
    auto vec = [&](Tensor& t, size_t n) -> int {
        if ((size_t)numel != n) return fail(RZ_ERR_INVALID, std::string("bad numel for ") + rest);
        RZ_HIP(m->upload(t, data, n, false));
        t.loaded = true;
        return 0;
    };
    auto mat = [&](Tensor& t, size_t r, size_t c) -> int {
        if ((size_t)numel != r * c) return fail(RZ_ERR_INVALID, std::string("bad numel for ") + rest);
        RZ_HIP(m->upload(t, data, r * c, true));
        t.loaded = true;
        return 0;
    };
    // host copies for the fused-LayerNorm packing (fold_block): kept for the life of the handle (16.5 MB per block), so that a
    // partial reload (load_state_dict(strict=False), one rz_load_weight) re-folds from complete data; only the tensors that enter
    // c1 / c2 (norm1 / norm2, q|k|v, fc1) invalidate the fold
    const bool keep = m->dt != RZ_F32;
    // the host copy follows a SUCCESSFUL upload only: a refused tensor (wrong numel) must not truncate it (found by tools/host_asan.sh)
    auto kept = [&](std::vector<float>& hv, int rc) { if (rc == 0 && keep) { hv.assign(data, data + numel); b.folded = false; } return rc; };
    if (!strcmp(rest, "norm1.weight")) return kept(b.h_g1, vec(b.ln1_g, D));
    if (!strcmp(rest, "norm1.bias")) return kept(b.h_be1, vec(b.ln1_b, D));
    if (!strcmp(rest, "norm2.weight")) return kept(b.h_g2, vec(b.ln2_g, D));
    if (!strcmp(rest, "norm2.bias")) return kept(b.h_be2, vec(b.ln2_b, D));
    if (!strcmp(rest, "layer_scale1.lambda1")) return vec(b.ls1, D);
    if (!strcmp(rest, "layer_scale2.lambda1")) return vec(b.ls2, D);
    if (!strcmp(rest, "attention.output.dense.weight")) return mat(b.wo, D, D);
    if (!strcmp(rest, "attention.output.dense.bias")) return vec(b.bo, D);
    if (!strcmp(rest, "mlp.fc1.weight")) return kept(b.h_w1, mat(b.w1, F, D));
    if (!strcmp(rest, "mlp.fc1.bias")) return kept(b.h_b1, vec(b.b1, F));
    if (!strcmp(rest, "mlp.fc2.weight")) return mat(b.w2, D, F);
    if (!strcmp(rest, "mlp.fc2.bias")) return vec(b.b2, D);
    const char* names[3] = {"query", "key", "value"};
    for (int i = 0; i < 3; ++i) {
        char wn[64], bn[64];
        snprintf(wn, sizeof wn, "attention.attention.%s.weight", names[i]);
        snprintf(bn, sizeof bn, "attention.attention.%s.bias", names[i]);
        if (!strcmp(rest, wn)) {
            if ((size_t)numel != D * D) return fail(RZ_ERR_INVALID, "bad numel for qkv weight");
            int rc = put_rows(m, b.wqkv, 3 * D, D, i * D, data, D, i == 0 ? qscale : 1.f, true);
            if (rc) return rc;
            if (keep) {
                b.h_wqkv.resize(3 * D * D);
                for (size_t e = 0; e < D * D; ++e) b.h_wqkv[i * D * D + e] = data[e] * (i == 0 ? qscale : 1.f);
                b.folded = false;
            }
            b.qkv_parts |= (1 << i);
            return 0;
        }
        if (!strcmp(rest, bn)) {
            if ((size_t)numel != D) return fail(RZ_ERR_INVALID, "bad numel for qkv bias");
            int rc = put_rows(m, b.bqkv, 3, D, i, data, 1, i == 0 ? qscale : 1.f, false);
            if (rc) return rc;
            if (keep) {
                b.h_bqkv.resize(3 * D);
                for (size_t e = 0; e < D; ++e) b.h_bqkv[i * D + e] = data[e] * (i == 0 ? qscale : 1.f);
                b.folded = false;
            }
            b.qkv_parts |= (8 << i);
            return 0;
        }
    }
    return fail(RZ_ERR_INVALID, std::string("unknown Dinov2Layer tensor: ") + rest);
}

int load_text_layer(rz_model* m, TextLayer& l, const char* rest, const float* data, int64_t numel) {
    const size_t D = m->D, F = m->cfg.text_intermediate_size;
    const float qscale = 1.0f / sqrtf((float)(m->D / m->H));
    auto vec = [&](Tensor& t, size_t n) -> int {
        if ((size_t)numel != n) return fail(RZ_ERR_INVALID, std::string("bad numel for ") + rest);
        RZ_HIP(m->upload(t, data, n, false));
        t.loaded = true;
        return 0;
    };
    auto mat = [&](Tensor& t, size_t r, size_t c) -> int {
        if ((size_t)numel != r * c) return fail(RZ_ERR_INVALID, std::string("bad numel for ") + rest);
        RZ_HIP(m->upload(t, data, r * c, true));
        t.loaded = true;
        return 0;
    };
    if (!strcmp(rest, "attention.attn.o.weight")) return mat(l.wo, D, D);
    if (!strcmp(rest, "attention.attn.o.bias")) return vec(l.bo, D);
    if (!strcmp(rest, "attention.LayerNorm.weight")) return vec(l.lna_g, D);
    if (!strcmp(rest, "attention.LayerNorm.bias")) return vec(l.lna_b, D);
    if (!strcmp(rest, "intermediate.dense.weight")) return mat(l.w1, F, D);
    if (!strcmp(rest, "intermediate.dense.bias")) return vec(l.b1, F);
    if (!strcmp(rest, "output.dense.weight")) return mat(l.w2, D, F);
    if (!strcmp(rest, "output.dense.bias")) return vec(l.b2, D);
    if (!strcmp(rest, "output.LayerNorm.weight")) return vec(l.lno_g, D);
    if (!strcmp(rest, "output.LayerNorm.bias")) return vec(l.lno_b, D);
    const char* names[3] = {"q", "k", "v"};
    for (int i = 0; i < 3; ++i) {
        char wn[64], bn[64];
        snprintf(wn, sizeof wn, "attention.attn.%s.weight", names[i]);
        snprintf(bn, sizeof bn, "attention.attn.%s.bias", names[i]);
        if (!strcmp(rest, wn)) {
            if ((size_t)numel != D * D) return fail(RZ_ERR_INVALID, "bad numel for qkv weight");
            int rc = put_rows(m, l.wqkv, 3 * D, D, i * D, data, D, i == 0 ? qscale : 1.f, true);
            if (rc) return rc;
            l.qkv_parts |= (1 << i);
            return 0;
        }
        if (!strcmp(rest, bn)) {
            if ((size_t)numel != D) return fail(RZ_ERR_INVALID, "bad numel for qkv bias");
            int rc = put_rows(m, l.bqkv, 3, D, i, data, 1, i == 0 ? qscale : 1.f, false);
            if (rc) return rc;
            l.qkv_parts |= (8 << i);
            return 0;
        }
    }
    return fail(RZ_ERR_INVALID, std::string("unknown MPNetLayer tensor: ") + rest);
}

// fp32 mode, vision encoder: the GEMM runs on the f16 matrix pipe over hi/lo-split operands (gemm.hip, launch_gemm_split_f32out) when
// the weight has a split copy and A lies in one of the vision workspaces (its token row picks the slice of the split scratch).
// reuse_split: A was split by the previous call (q|k and v projections share their input).
enum { A_F32 = 0, A_SPLIT = 1, A_REUSE = 2 };   // the A operand: fp32 (split here), already [M][3K] hi|lo|hi planes, or split by the previous call

// mx: 0 = three f16 planes; else the MX form, and bits 1 / 2 say which attention operands leave with an e4m3 pair plane instead of an f16 lo plane
// (2: V^T, 4: q | k)
int gemm_f32_split(rz_model* m, int epi, GemmArgs g, int a_mode, bool out_split, hipStream_t s, bool* done, int mx = 0) {
    *done = false;
    if (!(epi == EPI_HEADS || epi == EPI_VT || epi == EPI_GELU || epi == EPI_RESID_SCALE || epi == EPI_PATCH)) return 0;
    if (g.M % 128 || g.lda != g.K || g.ldw != g.K) return 0;
    const char* w3 = nullptr;
    for (const auto& e : m->split_w) {
        const char* w = (const char*)g.W;
        if (w >= e.p && w < e.p + e.bytes) {
            const size_t row = (size_t)(w - e.p) / ((size_t)g.K * 4);
            if (e.p + row * g.K * 4 != w) return 0;
            w3 = mx ? e.p4 + row * 4 * g.K : e.p3 + row * 3 * g.K * 2;
            g.mx_w_e8_hi = e.e8_hi; g.mx_w_e8_lo = e.e8_hi - 11;
            break;
        }
    }
    if (!w3) return mx ? fail(RZ_ERR_STATE, "MX GEMM requested for a weight without an MX copy") : 0;
    if (mx && m->o_f32_drop() != 0) return fail(RZ_ERR_STATE, "f32_drop (accuracy ablation) runs on the three-plane form: set gemm_f32_mx = 0");
    if (mx) {
        // MX form (rz_common.h): rows of 4 K bytes = 2 K f16-element units; A_SPLIT = already in that form (LayerNorm / attention / fc1
        // epilogue wrote it), A_F32 = split here (the patch embedding's im2col matrix)
        if (a_mode == A_REUSE) return fail(RZ_ERR_STATE, "MX GEMM: A_REUSE is not used on this path");
        if (a_mode == A_F32) {
            if (!m->asplit.p || (size_t)g.M * 4 * g.K > m->asplit.bytes) return fail(RZ_ERR_STATE, "MX GEMM: split scratch too small");
            RZ_HIP(launch_split3((const float*)g.A, g.lda, m->asplit.p, g.M, g.K, 2, g.ovf_flag, s));
            g.A = m->asplit.p;
        }
        g.W = w3; g.lda = g.ldw = 2 * (int64_t)g.K; g.K = 2 * g.K;
        if (!gemm_v7_mx_ok(g)) return fail(RZ_ERR_STATE, "MX GEMM: shape not supported");
        const int out_kind = (epi == EPI_RESID_SCALE || epi == EPI_PATCH) ? 0 : epi == EPI_GELU ? 2 :
                             ((epi == EPI_VT && (mx & 2)) || (epi == EPI_HEADS && (mx & 4))) ? 3 : 1;       // 3: hi f16 + e4m3 pair planes (MX attention)
        if ((out_kind != 0) != out_split) return fail(RZ_ERR_STATE, "MX GEMM: output form mismatch");
        g.variant = m->o_gemm_variant(); g.small_tile = m->o_gemm_small_tile();
        // which kernel: 0 auto — the 128 x 128 kernel where the 16-bit kernels' cost model prefers small tiles (round 6: one 1024^2 image is 63 big tiles for
        // an N = 768 GEMM on 256 CUs), else the persistent loop; 1 / 7 / 8 force the 128 x 128 / one-tile-per-workgroup / persistent kernel (same bits)
        const bool small_ok = gemm_small_mx_ok(epi, out_kind, g);
        if (small_ok && (g.variant == 1 || (g.variant == 0 && gemm_small_mx_pays(epi, g)))) RZ_HIP(launch_gemm_small_mx(epi, g, out_kind, s));
        else if (g.variant != 7 && gemm_v8_mx_ok(epi, out_kind, g)) RZ_HIP(launch_gemm_v8_mx(epi, out_kind, g, s));       // persistent loop: the epilogue's stores under the next tile's K loop
        else RZ_HIP(launch_gemm_v7_mx(epi, g, out_kind, s));
        *done = true;
        return 0;
    }
    if (a_mode != A_SPLIT) {
        if (!m->asplit.p) return 0;
        const size_t maxk = std::max((size_t)m->F, (size_t)m->KPAD);
        const char* a = (const char*)g.A;
        auto row_in = [&](const DevBuf& b, size_t row_bytes, size_t* row) {
            if (!b.p || a < (const char*)b.p || a >= (const char*)b.p + b.bytes) return false;
            *row = (size_t)(a - (const char*)b.p) / row_bytes;
            return true;
        };
        size_t row = 0;
        if (!(row_in(m->xn, (size_t)m->D * 6, &row) || row_in(m->ctx, (size_t)m->D * 6, &row) || row_in(m->mid, maxk * 6, &row))) return 0;
        char* a3 = (char*)m->asplit.p + row * 3 * maxk * 2;
        if (a3 + (size_t)g.M * 3 * g.K * 2 > (char*)m->asplit.p + m->asplit.bytes) return 0;
        if (a_mode == A_F32) RZ_HIP(launch_split3((const float*)g.A, g.lda, a3, g.M, g.K, 0, g.ovf_flag, s));
        g.A = a3;
    }
    // accuracy ablation (option f32_drop): this product class on the hi planes alone = the first K of the 3 K columns
    const int cls = epi == EPI_HEADS ? 1 : epi == EPI_VT ? 2 : epi == EPI_GELU ? 8 : epi == EPI_PATCH ? 32 : (g.K == m->F ? 16 : 4);
    const bool hi_only = (m->o_f32_drop() & cls) != 0;
    g.W = w3; g.lda = g.ldw = 3 * (int64_t)g.K; g.K = hi_only ? g.K : 3 * g.K;
    RZ_HIP(launch_gemm_split_f32out(epi, g, s, out_split));
    *done = true;
    return 0;
}

// fp32 mode, split forms: a block's q|k and v projections (A = the LayerNorm's planes, both outputs planes for the split attention) as ONE launch of the 128 x 128 family
// where each would have taken that family alone (gemm.hip launch_gemm_pair_f32); *done = false: not applicable, the caller launches the two
int gemm_f32_split_qkv_pair(rz_model* m, const void* xn, const DinoBlock& b, int M, int np, void* qk, void* vt, int mx, hipStream_t s, bool* done) {
    *done = false;
    const int D = m->D, H = m->H;
    if (!m->o_gemm_qkv_pair() || m->o_f32_drop() != 0 || (mx & 4) || M % 128) return 0;
    const rz_model::SplitW* e = nullptr;
    for (const auto& c : m->split_w)
        if ((const char*)b.wqkv.p == c.p) { e = &c; break; }
    if (!e || (mx && !e->p4)) return 0;
    GemmArgs ga, gb;
    ga.A = xn; ga.M = M; ga.N = 2 * D; ga.bias = (const float*)b.bqkv.p; ga.out = qk; ga.ldo = 0; ga.scale = nullptr; ga.resid = nullptr; ga.ldr = 0;
    ga.rows_per_image = np; ga.heads_total = 2 * H; ga.plane_off = (int64_t)M * 2 * D; ga.ovf_flag = (unsigned*)m->ovf.p; ga.run_if = m->run_if;
    ga.variant = m->o_gemm_variant(); ga.small_tile = m->o_gemm_small_tile(); ga.raster = m->o_gemm_raster();
    ga.mx_w_e8_hi = e->e8_hi; ga.mx_w_e8_lo = e->e8_hi - 11;
    const int form = mx ? 1 : 0, v_kind = (mx & 2) ? 3 : 1;
    if (mx) { ga.W = e->p4; ga.lda = ga.ldw = 2 * (int64_t)D; ga.K = 2 * D; }
    else { ga.W = e->p3; ga.lda = ga.ldw = 3 * (int64_t)D; ga.K = 3 * D; }
    gb = ga;
    gb.W = (const char*)ga.W + (size_t)2 * D * (mx ? 4 : 6) * D;      // row 2 D of the split copy: 4 K (MX) / 6 K (three planes) bytes per row
    gb.N = D; gb.bias = (const float*)b.bqkv.p + 2 * D; gb.out = vt; gb.heads_total = H; gb.plane_off = (int64_t)M * D;
    if (pick(m->opt.gemm_qkv_pair, g_opt.gemm_qkv_pair) == 2 && ga.variant == 0) ga.variant = gb.variant = 1;      // A/B: the pair wherever the shapes allow
    if (!gemm_pair_f32_ok(form, ga, gb, v_kind)) return 0;
    ProfScope ps(m, RZ_PROF_GEMM, s);
    RZ_HIP(launch_gemm_pair_f32(form, ga, gb, v_kind, s));
    *done = true;
    return 0;
}

// fp32 mode, text encoder (round 6): the MPNet GEMMs (EPI_STORE q|k|v, EPI_RESID_ADD o / fc2, EPI_GELU fc1) on the three-plane f16 form —
// the weight's [hi | hi | lo] copy was built by rz_weights_ready; fp32 outputs, or planes again for fc1 (exact-erf GELU).
// Plane overflows raise the TEXT guard word (ovf[5]): rz_text_forward repeats the encode on the exact kernels behind a predicate, as the vision forward does.
// A is ALREADY in planes (its producer wrote them: the LayerNorm, the attention, fc1's GELU epilogue) inside `tasplit`; out_split: EPI_GELU writes planes too.
int gemm_text_split(rz_model* m, int epi, GemmArgs g, hipStream_t s, bool out_split) {
    const char* w3 = nullptr;
    for (const auto& e : m->split_w)
        if ((const char*)g.W == e.p && !e.p4) { w3 = e.p3; break; }
    if (!w3 || g.M % 128) return fail(RZ_ERR_STATE, "text GEMM: no three-plane copy of this weight");
    g.ovf_flag = (unsigned*)m->ovf.p + 5;
    g.W = w3; g.lda = g.ldw = 3 * (int64_t)g.K; g.K = 3 * g.K;
    ProfScope ps(m, RZ_PROF_GEMM, s);
    RZ_HIP(launch_gemm_split_f32out(epi, g, s, out_split));
    return 0;
}

// a_mode / out_split / plane_off: fp32 mode's hi/lo-split path only (see gemm_f32_split); with A_SPLIT or out_split the call MUST take it
int gemm(rz_model* m, int epi, const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K, const float* bias,
         void* out, int64_t ldo, const float* scale, float* resid, int64_t ldr, int rpi, int heads, hipStream_t s, int a_mode = A_F32,
         bool out_split = false, int64_t plane_off = 0, int mx = 0) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out = out; g.ldo = ldo;
    g.scale = scale; g.resid = resid; g.ldr = ldr; g.rows_per_image = rpi; g.heads_total = heads; g.plane_off = plane_off;
    g.variant = m->o_gemm_variant(); g.small_tile = m->o_gemm_small_tile(); g.raster = m->o_gemm_raster(); g.ovf_flag = (unsigned*)m->ovf.p; g.run_if = m->run_if;
    ProfScope ps(m, RZ_PROF_GEMM, s);
    if (m->dt == RZ_F32 && m->o_gemm_f32_split()) {
        bool done = false;
        int rc = gemm_f32_split(m, epi, g, a_mode, out_split, s, &done, mx);
        if (rc || done) return rc;
    }
    if (a_mode == A_SPLIT || out_split) return fail(RZ_ERR_STATE, "split GEMM requested but not applicable");
    RZ_HIP(launch_gemm(m->dt, epi, g, s));
    return 0;
}

float t_to_f32(int dt, uint16_t v) {
    if (dt == RZ_BF16) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
    _Float16 h; memcpy(&h, &v, 2); return (float)h;
}

// c1[n] = sum_k gamma[k] T(W[n][k]) (over the weights as the MFMAs see them, so that LN(x) W^T = rstd ((x gamma) W^T - mu c1) + c2
// holds exactly for them), c2 = W beta + b.
int fold_matrix(rz_model* m, const std::vector<float>& W, const std::vector<float>& bias, const std::vector<float>& gamma,
                const std::vector<float>& beta, size_t N, size_t K, Tensor& c1, Tensor& c2) {
    if (W.size() != N * K || bias.size() != N || gamma.size() != K || beta.size() != K)
        return fail(RZ_ERR_STATE, "fused LayerNorm packing: host copies of the block's weights are incomplete (W " + std::to_string(W.size()) + " of " +
                                      std::to_string(N * K) + ", bias " + std::to_string(bias.size()) + " of " + std::to_string(N) + ", gamma " +
                                      std::to_string(gamma.size()) + ", beta " + std::to_string(beta.size()) + " of " + std::to_string(K) + ")");
    std::vector<float> v1(N), v2(N);
    for (size_t n = 0; n < N; ++n) {
        double s1 = 0.0, s2 = 0.0;
        for (size_t k = 0; k < K; ++k) {
            const float w = W[n * K + k];
            const float wr = t_to_f32(m->dt, m->dt == RZ_BF16 ? f32_to_bf16(w) : f32_to_f16(w));
            s1 += (double)wr * (double)gamma[k];
            s2 += (double)w * (double)beta[k];
        }
        v1[n] = (float)s1;
        v2[n] = (float)(s2 + (double)bias[n]);
    }
    RZ_HIP(m->upload(c1, v1.data(), N, false));
    RZ_HIP(m->upload(c2, v2.data(), N, false));
    c1.loaded = c2.loaded = true;
    return 0;
}

int fold_block(rz_model* m, DinoBlock& b) {
    if (b.folded) return 0;
    if (b.qkv_parts != 63) return fail(RZ_ERR_STATE, "fused LayerNorm packing: q|k|v pieces of the block are incomplete");
    const size_t D = m->D, F = m->F;
    int rc;
    if ((rc = fold_matrix(m, b.h_wqkv, b.h_bqkv, b.h_g1, b.h_be1, 3 * D, D, b.c1qkv, b.c2qkv))) return rc;
    if ((rc = fold_matrix(m, b.h_w1, b.h_b1, b.h_g2, b.h_be2, F, D, b.c1_1, b.c2_1))) return rc;
    b.folded = true;
    return 0;
}

// GEMM that follows a LayerNorm, fused form: A = un-normalised residual copy, per-row (mean, rstd) in `stat`
int gemm_ln(rz_model* m, int epi, const void* hb, const Tensor& wf, const Tensor& c1, const Tensor& c2, const float* stat, int M, int N,
            int np, void* out, int64_t ldo, int heads, void* out2, int heads2, int split_n, hipStream_t s) {
    GemmArgs g;
    g.A = hb; g.lda = m->D; g.W = wf.p; g.ldw = m->D; g.M = M; g.N = N; g.K = m->D; g.bias = (const float*)c2.p; g.out = out; g.ldo = ldo;
    g.scale = (const float*)c1.p; g.resid = nullptr; g.ldr = 0; g.rows_per_image = np; g.heads_total = heads;
    g.out2 = out2; g.heads_total2 = heads2; g.split_n = split_n; g.ln_stat = stat; g.variant = m->o_gemm_variant(); g.small_tile = m->o_gemm_small_tile(); g.raster = m->o_gemm_raster();
    ProfScope ps(m, RZ_PROF_GEMM, s);
    RZ_HIP(launch_gemm(m->dt, epi, g, s));
    return 0;
}

// the block's q|k and v projections behind a fused LayerNorm as ONE launch of the 128 x 128 family (gemm.hip gemm_pair_kernel); *done = false: not applicable, the caller launches the two
int gemm_ln_qkv_pair(rz_model* m, const void* hb, const DinoBlock& b, const float* stat, int M, int np, void* qk, void* vt, hipStream_t s, bool* done) {
    const int D = m->D, H = m->H;
    *done = false;
    if (!m->o_gemm_qkv_pair()) return 0;
    GemmArgs ga, gb;
    ga.A = hb; ga.lda = D; ga.W = b.wqkv.p; ga.ldw = D; ga.M = M; ga.N = 2 * D; ga.K = D; ga.bias = (const float*)b.c2qkv.p; ga.out = qk; ga.ldo = 0;
    ga.scale = (const float*)b.c1qkv.p; ga.resid = nullptr; ga.ldr = 0; ga.rows_per_image = np; ga.heads_total = 2 * H; ga.ln_stat = stat;
    ga.variant = m->o_gemm_variant(); ga.small_tile = m->o_gemm_small_tile(); ga.raster = m->o_gemm_raster();
    gb = ga;
    gb.W = (const char*)b.wqkv.p + (size_t)2 * D * D * dsize(m->dt); gb.N = D; gb.bias = (const float*)b.c2qkv.p + 2 * D; gb.scale = (const float*)b.c1qkv.p + 2 * D;
    gb.out = vt; gb.heads_total = H;
    if (!gemm_pair_ok(m->dt, EPI_HEADS_LN, ga, EPI_VT_LN, gb)) return 0;
    ProfScope ps(m, RZ_PROF_GEMM, s);
    RZ_HIP(launch_gemm_pair(m->dt, EPI_HEADS_LN, ga, EPI_VT_LN, gb, s));
    *done = true;
    return 0;
}

// residual GEMM that precedes a LayerNorm, fused form: also writes the T copy of the new residual and partial statistics,
// then the 12 partials per row are merged into (mean, rstd)
int gemm_resid_ln(rz_model* m, const void* A, int64_t lda, const Tensor& W, const Tensor& bias, const Tensor& ls, int M, int K, float* h,
                  int np, const Tensor& next_gamma, void* hb, float* part, float* mu, float* stat, float eps, hipStream_t s) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W.p; g.ldw = K; g.M = M; g.N = m->D; g.K = K; g.bias = (const float*)bias.p; g.out = nullptr; g.ldo = 0;
    g.scale = (const float*)ls.p; g.resid = h; g.ldr = m->D; g.rows_per_image = np; g.heads_total = 0; g.ln_part = part; g.ln_hb = hb; g.ln_gamma = (const float*)next_gamma.p; g.ln_mu = mu;
    g.variant = m->o_gemm_variant(); g.small_tile = m->o_gemm_small_tile(); g.raster = m->o_gemm_raster();
    {
        ProfScope ps(m, RZ_PROF_GEMM, s);
        RZ_HIP(launch_gemm(m->dt, EPI_RESID_SCALE_LN, g, s));
    }
    ProfScope ps(m, RZ_PROF_ROWOPS, s);
    RZ_HIP(launch_ln_finalize(part, mu, stat, eps, M, s));
    return 0;
}

int gemm_qkv(rz_model* m, const void* xn, const DinoBlock& b, int M, int np, void* qk, void* vt, hipStream_t s) {
    const int D = m->D, H = m->H;
    GemmArgs g;
    g.A = xn; g.lda = D; g.W = b.wqkv.p; g.ldw = D; g.M = M; g.N = 3 * D; g.K = D; g.bias = (const float*)b.bqkv.p;
    g.out = qk; g.ldo = 0; g.scale = nullptr; g.resid = nullptr; g.ldr = 0; g.rows_per_image = np; g.heads_total = 2 * H;
    g.out2 = vt; g.heads_total2 = H; g.split_n = 2 * D; g.variant = m->o_gemm_variant(); g.small_tile = m->o_gemm_small_tile(); g.raster = m->o_gemm_raster();
    const char* wv = (const char*)b.wqkv.p + (size_t)2 * D * D * dsize(m->dt);
    int rc;
    auto pair = [&](bool* done) -> int {      // 16-bit modes: the q|k and v launches as one (gemm.hip gemm_pair_kernel)
        *done = false;
        if (m->dt == RZ_F32 || !m->o_gemm_qkv_pair()) return 0;
        GemmArgs ga = g, gb;
        ga.N = 2 * D; ga.out2 = nullptr; ga.heads_total2 = 0; ga.split_n = 0;
        gb = ga;
        gb.W = wv; gb.N = D; gb.bias = (const float*)b.bqkv.p + 2 * D; gb.out = vt; gb.heads_total = H;
        if (!gemm_pair_ok(m->dt, EPI_HEADS, ga, EPI_VT, gb)) return 0;
        ProfScope ps(m, RZ_PROF_GEMM, s);
        RZ_HIP(launch_gemm_pair(m->dt, EPI_HEADS, ga, EPI_VT, gb, s));
        *done = true;
        return 0;
    };
    bool done = false;
    if (m->o_gemm_qkv_pair_first(M) && ((rc = pair(&done)) || done)) return rc;
    if (gemm_qkv_fused_ok(m->dt, g)) {
        ProfScope ps(m, RZ_PROF_GEMM, s);
        RZ_HIP(launch_gemm(m->dt, EPI_QKV, g, s));
        return 0;
    }
    if ((rc = pair(&done)) || done) return rc;
    if ((rc = gemm(m, EPI_HEADS, xn, D, b.wqkv.p, D, M, 2 * D, D, (const float*)b.bqkv.p, qk, 0, nullptr, nullptr, 0, np, 2 * H, s))) return rc;
    return gemm(m, EPI_VT, xn, D, wv, D, M, D, D, (const float*)b.bqkv.p + 2 * D, vt, 0, nullptr, nullptr, 0, np, H, s, A_REUSE);
}

}  // namespace

extern "C" {

const char* rz_last_error(void) { return g_err.c_str(); }
const char* rz_version(void) { return "radzero_hip 0.1 (gfx950)"; }

int rz_create(const rz_config* cfg, rz_handle_t* out) {
    if (!cfg || !out) return fail(RZ_ERR_INVALID, "rz_create: null argument");
    if (cfg->compute_dtype < 0 || cfg->compute_dtype > 2) return fail(RZ_ERR_INVALID, "rz_create: compute_dtype");
    if (cfg->hidden_size != 768 || cfg->num_attention_heads != 12)
        return fail(RZ_ERR_UNSUPPORTED, "rz_create: kernels are specialised for hidden 768 / 12 heads of 64 (dinov2-base, mpnet-base)");
    if (cfg->patch_size <= 0 || cfg->num_channels <= 0 || cfg->vit_layers < 0 || cfg->align_layers < 0 || cfg->text_layers < 0)
        return fail(RZ_ERR_INVALID, "rz_create: layer counts / patch size");
    if ((cfg->hidden_size * cfg->mlp_ratio) % 128 || cfg->text_intermediate_size % 128)
        return fail(RZ_ERR_INVALID, "rz_create: intermediate sizes must be multiples of 128");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(RZ_ERR_STATE, "rz_create: no HIP device visible (this library has no CPU fallback)");
    rz_model* m = new rz_model();
    m->cfg = *cfg;
    m->dt = cfg->compute_dtype;
    m->D = cfg->hidden_size;
    m->H = cfg->num_attention_heads;
    m->F = cfg->hidden_size * cfg->mlp_ratio;
    m->KP = cfg->num_channels * cfg->patch_size * cfg->patch_size;
    m->KPAD = round_up(m->KP, 64);
    m->blocks.resize(cfg->vit_layers + cfg->align_layers);
    m->tlayers.resize(cfg->text_layers);
    *out = m;
    return 0;
}

int rz_destroy(rz_handle_t m) {
    if (!m) return 0;
    (void)hipDeviceSynchronize();        // nothing of this handle may still be in flight when its buffers go away
    for (void* p : m->allocs) (void)hipFree(p);
    for (auto& kv : m->pos_tables) kv.second.buf.release();
    if (m->ovf_host) (void)hipHostFree(m->ovf_host);
    for (auto& e : m->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (m->side_ready) {
        for (int i = 0; i < 2; ++i) { (void)hipStreamDestroy(m->side[i]); (void)hipEventDestroy(m->side_done[i]); }
        (void)hipEventDestroy(m->fork);
    }
    delete m;
    return 0;
}

int rz_load_weight(rz_handle_t m, const char* name, const float* data, int64_t numel) {
    if (!m || !name || !data || numel <= 0) return fail(RZ_ERR_INVALID, "rz_load_weight: bad argument");
    m->split_dirty = true;
    const size_t D = m->D;
    int idx;
    const char* rest;
    auto vec = [&](Tensor& t, size_t n) -> int {
        if ((size_t)numel != n) return fail(RZ_ERR_INVALID, std::string("bad numel for ") + name);
        RZ_HIP(m->upload(t, data, n, false));
        t.loaded = true;
        return 0;
    };
    if (parse_layer(name, "vision_model.encoder.layer.", &idx, &rest)) {
        if (idx < 0 || idx >= m->cfg.vit_layers) return fail(RZ_ERR_INVALID, std::string("layer index out of range: ") + name);
        return load_dino_block(m, m->blocks[idx], rest, data, numel);
    }
    if (parse_layer(name, "align_transformer.transformer_layers.layer.", &idx, &rest)) {
        if (idx < 0 || idx >= m->cfg.align_layers) return fail(RZ_ERR_INVALID, std::string("layer index out of range: ") + name);
        return load_dino_block(m, m->blocks[m->cfg.vit_layers + idx], rest, data, numel);
    }
    if (parse_layer(name, "text_model.encoder.layer.", &idx, &rest)) {
        if (idx < 0 || idx >= m->cfg.text_layers) return fail(RZ_ERR_INVALID, std::string("layer index out of range: ") + name);
        return load_text_layer(m, m->tlayers[idx], rest, data, numel);
    }
    if (!strcmp(name, "vision_model.embeddings.cls_token")) {
        if ((size_t)numel != D) return fail(RZ_ERR_INVALID, "bad numel for cls_token");
        m->cls_host.assign(data, data + D);
        m->cls_loaded = true;
        return 0;
    }
    if (!strcmp(name, "vision_model.embeddings.patch_embeddings.projection.bias")) {
        if ((size_t)numel != D) return fail(RZ_ERR_INVALID, "bad numel for patch bias");
        m->patch_bias_host.assign(data, data + D);
        m->patch_bias_loaded = true;
        return 0;
    }
    if (!strcmp(name, "vision_model.embeddings.patch_embeddings.projection.weight")) {
        if ((size_t)numel != D * (size_t)m->KP) return fail(RZ_ERR_INVALID, "bad numel for patch weight");
        std::vector<float> padded(D * (size_t)m->KPAD, 0.f);          // [768][588] -> [768][640], zero padded
        for (size_t r = 0; r < D; ++r) memcpy(&padded[r * m->KPAD], data + r * m->KP, m->KP * sizeof(float));
        RZ_HIP(m->upload(m->patch_w, padded.data(), padded.size(), true));
        m->patch_w.loaded = true;
        return 0;
    }
    if (!strcmp(name, "vision_model.layernorm.weight")) return vec(m->vit_ln_g, D);
    if (!strcmp(name, "vision_model.layernorm.bias")) return vec(m->vit_ln_b, D);
    if (!strcmp(name, "text_model.embeddings.word_embeddings.weight")) return vec(m->word_emb, (size_t)m->cfg.vocab_size * D);
    if (!strcmp(name, "text_model.embeddings.position_embeddings.weight")) return vec(m->pos_emb, (size_t)m->cfg.max_position_embeddings * D);
    if (!strcmp(name, "text_model.embeddings.LayerNorm.weight")) return vec(m->temb_ln_g, D);
    if (!strcmp(name, "text_model.embeddings.LayerNorm.bias")) return vec(m->temb_ln_b, D);
    if (!strcmp(name, "loss_fns.RadZeroLoss.layer_norm.weight")) return vec(m->shared_ln_g, D);
    if (!strcmp(name, "loss_fns.RadZeroLoss.layer_norm.bias")) return vec(m->shared_ln_b, D);
    if (!strcmp(name, "loss_fns.RadZeroLoss.loss_temperature")) {
        if (numel != 1) return fail(RZ_ERR_INVALID, "bad numel for loss_temperature");
        m->tau = expf(data[0]);       // stored as log(tau) (losses.py:54-56); tau = exp(param) (losses.py:177-181)
        m->tau_loaded = true;
        m->tau_session = true;
        return 0;
    }
    if (!strcmp(name, "loss_fns.RadZeroLoss.attn_temperature")) {      // only present when the config sets one (losses.py:57-63)
        if (numel != 1) return fail(RZ_ERR_INVALID, "bad numel for attn_temperature");
        m->attn_tau = expf(data[0]);
        m->attn_tau_loaded = true;
        m->attn_tau_session = true;
        return 0;
    }
    // tensors the path never reads: position_embeddings (interpolated on the host and passed through
    // rz_set_position_table), mask_token, pooler, relative_attention_bias (passed expanded to rz_text_forward)
    if (!strcmp(name, "vision_model.embeddings.position_embeddings") || !strcmp(name, "vision_model.embeddings.mask_token") ||
        !strncmp(name, "text_model.pooler.", 18) || !strcmp(name, "text_model.encoder.relative_attention_bias.weight"))
        return 0;
    return fail(RZ_ERR_INVALID, std::string("rz_load_weight: unknown tensor name: ") + name);
}

int rz_weights_ready(rz_handle_t m) {
    if (!m) return fail(RZ_ERR_INVALID, "null handle");
    auto need = [&](bool ok, const std::string& what) -> int { return ok ? 0 : fail(RZ_ERR_STATE, "weight not loaded: " + what); };
    int rc;
    if ((rc = need(m->cls_loaded, "vision_model.embeddings.cls_token"))) return rc;
    if ((rc = need(m->patch_bias_loaded && m->patch_w.loaded, "vision_model.embeddings.patch_embeddings.projection"))) return rc;
    if ((rc = need(m->vit_ln_g.loaded && m->vit_ln_b.loaded, "vision_model.layernorm"))) return rc;
    for (size_t i = 0; i < m->blocks.size(); ++i) {
        const DinoBlock& b = m->blocks[i];
        bool ok = b.ln1_g.loaded && b.ln1_b.loaded && b.qkv_parts == 63 && b.wo.loaded && b.bo.loaded && b.ls1.loaded &&
                  b.ln2_g.loaded && b.ln2_b.loaded && b.w1.loaded && b.b1.loaded && b.w2.loaded && b.b2.loaded && b.ls2.loaded;
        if ((rc = need(ok, "Dinov2Layer " + std::to_string(i)))) return rc;
    }
    if ((rc = need(m->word_emb.loaded && m->pos_emb.loaded && m->temb_ln_g.loaded && m->temb_ln_b.loaded, "text_model.embeddings"))) return rc;
    for (size_t i = 0; i < m->tlayers.size(); ++i) {
        const TextLayer& l = m->tlayers[i];
        bool ok = l.qkv_parts == 63 && l.wo.loaded && l.bo.loaded && l.lna_g.loaded && l.lna_b.loaded && l.w1.loaded &&
                  l.b1.loaded && l.w2.loaded && l.b2.loaded && l.lno_g.loaded && l.lno_b.loaded;
        if ((rc = need(ok, "MPNetLayer " + std::to_string(i)))) return rc;
    }
    if ((rc = need(m->shared_ln_g.loaded && m->shared_ln_b.loaded && m->tau_loaded, "loss_fns.RadZeroLoss"))) return rc;
    // a checkpoint that brought loss_temperature but no attn_temperature uses tau for the scores (losses.py:175-181): an attn_temperature
    // left over from an earlier checkpoint on this handle must not survive it
    if (m->tau_session && !m->attn_tau_session) m->attn_tau_loaded = false;
    m->tau_session = m->attn_tau_session = false;
    if (m->dt != RZ_F32)        // one-time packing of the fused-LayerNorm vectors (c1, c2): here, so that no forward call allocates
        for (auto& b : m->blocks)
            if ((rc = fold_block(m, b))) return rc;
    if (m->dt == RZ_F32 && m->split_dirty) {     // f16 [hi | hi | lo] copies of the vision encoder's matrices for the hi/lo-split GEMMs
        m->split_w.clear();
        // overflow guard words: [0] activations (cleared by every forward), [1] weights (checked here, once)
        // overflow guard words: [0] activations, [1] weights beyond the f16 range, [2] weights beyond their MX hi8 plane (cannot happen with per-matrix
        // scales unless a weight is not finite), [3] scratch of the per-matrix |w| maximum
        RZ_HIP(m->ovf.ensure(32, true));      // zeroed when allocated; a re-split (weights reloaded) clears the flags, not the re-run counter
        if (!m->ovf_host) RZ_HIP(hipHostMalloc((void**)&m->ovf_host, 16, hipHostMallocDefault));
        RZ_HIP(hipMemset(m->ovf.p, 0, 16));
        auto split = [&](Tensor& t, size_t N, size_t K) -> int {
            if (!t.p3) {
                RZ_HIP(hipMalloc(&t.p3, N * 3 * K * 2));
                m->allocs.push_back(t.p3);
            }
            RZ_HIP(launch_split3((const float*)t.p, (int64_t)K, t.p3, (int64_t)N, (int)K, 1, (unsigned*)m->ovf.p + 1, nullptr));
            if (!t.p4) {                 // the MX form beside it (K % 64 == 0 everywhere on this path: 640, 768, 3072)
                RZ_HIP(hipMalloc(&t.p4, N * 4 * K));
                m->allocs.push_back(t.p4);
            }
            // the MX copy's plane scales follow THIS matrix (round 5; rounds 1-4 fixed them at 2^-4 / 2^-15, and one matrix with |w| > 28 — the outlier
            // checkpoint's fc2 rows, 400 x the rest — sent the whole model back to the three-plane form): hi8 = e4m3(w / 2^e) with 2^e the smallest
            // power of two that brings the matrix's largest |w| inside e4m3's 448, lo8 = e4m3((w - f16(w)) / 2^(e - 11)).  Smaller weights keep
            // e4m3's relative precision down to 2^-6 of the scale (15 binades below the largest), and below that their correction terms are negligible.
            RZ_HIP(hipMemset((unsigned*)m->ovf.p + 3, 0, 4));
            RZ_HIP(launch_absmax_bits((const float*)t.p, (int64_t)(N * K), (unsigned*)m->ovf.p + 3, nullptr));
            unsigned bits = 0;
            RZ_HIP(hipMemcpy(&bits, (unsigned*)m->ovf.p + 3, 4, hipMemcpyDeviceToHost));
            float wmax;
            memcpy(&wmax, &bits, 4);
            int e8 = 123;                                       // 2^-4: the fixed scale of rounds 1-4 (an all-zero matrix)
            if (std::isfinite(wmax) && wmax > 0.f) {
                int ex = 0;
                const float fr = frexpf(wmax / 448.0f, &ex);           // wmax / 448 = fr 2^ex, fr in [0.5, 1)
                e8 = 127 + (fr == 0.5f ? ex - 1 : ex);                  // smallest 2^e >= wmax / 448
                e8 = std::min(std::max(e8, 127 - 40), 127 + 8);
            }
            RZ_HIP(launch_split3((const float*)t.p, (int64_t)K, t.p4, (int64_t)N, (int)K, 3, (unsigned*)m->ovf.p + 2, nullptr, e8));
            m->split_w.push_back({(const char*)t.p, N * K * 4, (const char*)t.p3, (const char*)t.p4, e8});
            return 0;
        };
        const size_t D = m->D, F = m->F;
        if ((rc = split(m->patch_w, D, m->KPAD))) return rc;
        for (auto& b : m->blocks) {
            if ((rc = split(b.wqkv, 3 * D, D)) || (rc = split(b.wo, D, D)) || (rc = split(b.w1, F, D)) || (rc = split(b.w2, D, F))) return rc;
        }
        // round 6: the text encoder's matrices in the three-plane form only (its GEMMs have 128-row operands: no MX form) — its exact-fp32 GEMMs were 4.8 ms of
        // a request's latency chain (72 launches of 56-224 us), longer than the whole vision forward of a 518^2 image
        auto split3_only = [&](Tensor& t, size_t N, size_t K) -> int {
            if (!t.p3) {
                RZ_HIP(hipMalloc(&t.p3, N * 3 * K * 2));
                m->allocs.push_back(t.p3);
            }
            RZ_HIP(launch_split3((const float*)t.p, (int64_t)K, t.p3, (int64_t)N, (int)K, 1, (unsigned*)m->ovf.p + 1, nullptr));
            m->split_w.push_back({(const char*)t.p, N * K * 4, (const char*)t.p3, nullptr, 123});
            return 0;
        };
        const size_t TFs = m->cfg.text_intermediate_size;
        for (auto& l : m->tlayers) {
            if ((rc = split3_only(l.wqkv, 3 * D, D)) || (rc = split3_only(l.wo, D, D)) || (rc = split3_only(l.w1, TFs, D)) || (rc = split3_only(l.w2, D, TFs))) return rc;
        }
        RZ_HIP(hipDeviceSynchronize());
        RZ_HIP(hipMemcpy(m->ovf_host, m->ovf.p, 16, hipMemcpyDeviceToHost));
        m->text_split_ok = !m->ovf_host[1] && !m->tlayers.empty();
        if (m->ovf_host[1]) m->split_w.clear();      // a weight beyond the f16 range: this checkpoint runs on the exact-fp32 GEMM kernels
        m->mx_weights_ok = !m->ovf_host[2];          // a weight beyond the hi8 plane's range: the three-plane f16 form only
        m->split_dirty = false;
    }
    return 0;
}

int rz_set_position_table(rz_handle_t m, int gh, int gw, const float* pos_host) {
    if (!m || !pos_host || gh <= 0 || gw <= 0) return fail(RZ_ERR_INVALID, "rz_set_position_table: bad argument");
    if (!m->cls_loaded || !m->patch_bias_loaded) return fail(RZ_ERR_STATE, "rz_set_position_table: load cls_token and patch bias first");
    const int D = m->D, nv = 1 + gh * gw, np = m->pad_tokens(nv);          // the largest padding any batch takes: rows beyond nv are zero
    std::vector<float> tbl((size_t)np * D, 0.f);
    for (int d = 0; d < D; ++d) tbl[d] = pos_host[d] + m->cls_host[d];
    for (int t = 1; t < nv; ++t)
        for (int d = 0; d < D; ++d) tbl[(size_t)t * D + d] = pos_host[(size_t)t * D + d] + m->patch_bias_host[d];
    auto& pt = m->pos_tables[{gh, gw}];
    RZ_HIP(hipDeviceSynchronize());
    RZ_HIP(pt.buf.ensure(tbl.size() * 4, false));
    RZ_HIP(hipMemcpy(pt.buf.p, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice));
    pt.n_valid = nv;
    pt.n_pad = np;
    return 0;
}

int rz_padded_tokens(rz_handle_t m, int n_tokens, int batch, int* n_pad_out) {
    if (!m || !n_pad_out || n_tokens <= 0 || batch < 0) return fail(RZ_ERR_INVALID, "rz_padded_tokens: bad argument");
    *n_pad_out = m->pad_tokens(n_tokens, batch);
    return 0;
}

int rz_reserve(rz_handle_t m, int max_batch, int max_tokens, int max_prompts, int max_len) {
    if (!m || max_batch < 0 || max_tokens < 0 || max_prompts < 0 || max_len < 0) return fail(RZ_ERR_INVALID, "rz_reserve: bad argument");
    const size_t es = dsize(m->dt), D = m->D, F = m->F;
    const int npad = m->pad_tokens(max_tokens);
    RZ_HIP(hipDeviceSynchronize());      // buffers may be re-allocated below: wait for any forward still using the old ones
    if (max_batch > 0 && max_tokens > 0) {
        const int B = std::max(max_batch, m->cap_batch), NP = std::max(npad, m->cap_npad);
        const size_t rows = (size_t)B * NP;
        RZ_HIP(m->h.ensure(rows * D * 4, true));
        const size_t ex = m->dt == RZ_F32 ? 6 : es;      // fp32 mode: room for the [hi | lo | hi] f16 planes of the hi/lo-split path
        RZ_HIP(m->xn.ensure(rows * D * ex, true));
        RZ_HIP(m->qk.ensure(rows * 2 * D * es, true));
        RZ_HIP(m->vt.ensure(rows * D * es, true));
        RZ_HIP(m->ctx.ensure(rows * D * ex, true));
        RZ_HIP(m->mid.ensure(rows * std::max(F, (size_t)m->KPAD) * ex, true));
        if (m->dt == RZ_F32) RZ_HIP(m->asplit.ensure(rows * 3 * std::max(F, (size_t)m->KPAD) * 2, false));
        if (m->dt != RZ_F32) {
            RZ_HIP(m->lnpart.ensure(rows * 24 * 4, true));
            RZ_HIP(m->lnstat.ensure(rows * 2 * 4, true));
            RZ_HIP(m->lnmu.ensure(rows * 4, true));
        }
        m->cap_batch = B;
        m->cap_npad = NP;
    }
    if (max_prompts > 0) {
        const int P = std::max(max_prompts, m->cap_prompts);
        RZ_HIP(m->qhat.ensure((size_t)P * D * 4, true));
        m->cap_prompts = P;
        if (max_len > 0) {
            const int trows = std::max(round_up(P * max_len, 128), m->cap_trows);
            const size_t TF = m->cfg.text_intermediate_size;
            RZ_HIP(m->th.ensure((size_t)trows * D * 4, true));
            RZ_HIP(m->tsum.ensure((size_t)trows * D * 4, true));
            RZ_HIP(m->txn.ensure((size_t)trows * D * es, true));
            RZ_HIP(m->tqkv.ensure((size_t)trows * 3 * D * es, true));
            RZ_HIP(m->tctx.ensure((size_t)trows * D * es, true));
            if (m->dt == RZ_F32) RZ_HIP(m->tasplit.ensure((size_t)trows * (6 * (size_t)D + 3 * TF) * 2, true));      // [hi | lo | hi] planes of txn, tctx, tmid
            RZ_HIP(m->tmid.ensure((size_t)trows * TF * es, true));
            m->cap_trows = trows;
        }
    }
    if (m->cap_batch > 0 && m->cap_prompts > 0)
        RZ_HIP(m->vws.ensure(vlcabs_workspace_floats(m->cap_batch, m->cap_prompts, m->cap_npad, m->D) * 4, false));
    return 0;
}

static int vision_forward_once(rz_handle_t m, const float* px, int B, int C, int Himg, int Wimg, float* tokens_out, void* stream);

int rz_vision_forward(rz_handle_t m, const float* px, int B, int C, int Himg, int Wimg, float* tokens_out, void* stream) {
    if (!m || !px) return fail(RZ_ERR_INVALID, "rz_vision_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    // fp32 mode on the f16 matrix pipe: the planes are f16 / e4m3, so an activation beyond a plane's range would turn into inf where fp32 stays
    // finite.  Every producer of planes raises a device word (rz_common.h flag_f16_range).  The guard never reads it on the host: behind the
    // first pass the SAME forward is enqueued once more on the exact-fp32 MFMA kernels with that word as every launch's predicate
    // (GemmArgs::run_if & co.: a wave reads the word and leaves at once when it is 0) — same stream, same buffers, nothing of the first pass
    // survives when it runs, ~75 empty launches when it does not.  So the call stays asynchronous, it can be captured (hipGraph replays carry the
    // guard), and the repeats are counted on the device (word [4], read lazily by rz_get_model_option "f32_split_guard_reruns").
    if (m->dt == RZ_F32) {       // builds the split weight copies and the guard words on first use
        const int rc0 = rz_weights_ready(m);
        if (rc0) return rc0;
    }
    const bool split = m->dt == RZ_F32 && (m->o_gemm_f32_split() || m->o_attn_f32_split()) && m->ovf.p;
    const bool guard = split && m->o_guard();
    // the word is cleared by a kernel, not a hipMemsetAsync: a memset node between kernels was not ordered with them on graph replay (see EPI_PATCH_LN below)
    if (split) RZ_HIP(launch_guard_word((unsigned*)m->ovf.p, 0, s));
    int rc = vision_forward_once(m, px, B, C, Himg, Wimg, tokens_out, stream);
    if (rc || !guard) return rc;
    m->force_exact = true;
    m->run_if = (const unsigned*)m->ovf.p;
    rc = vision_forward_once(m, px, B, C, Himg, Wimg, tokens_out, stream);
    m->force_exact = false;
    m->run_if = nullptr;
    if (rc) return rc;
    RZ_HIP(launch_guard_word((unsigned*)m->ovf.p, 1, s));
    return 0;
}

static int vision_forward_once(rz_handle_t m, const float* px, int B, int C, int Himg, int Wimg, float* tokens_out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (C != m->cfg.num_channels)   // TF:dinov2/modeling_dinov2.py:143-147
        return fail(RZ_ERR_INVALID, "Make sure that the channel dimension of the pixel values match with the one set in the configuration.");
    const int P = m->cfg.patch_size, gh = Himg / P, gw = Wimg / P;
    if (B <= 0 || gh <= 0 || gw <= 0) return fail(RZ_ERR_INVALID, "rz_vision_forward: image smaller than one patch or empty batch");
    int rc = rz_weights_ready(m);
    if (rc) return rc;
    auto it = m->pos_tables.find({gh, gw});
    if (it == m->pos_tables.end()) return fail(RZ_ERR_STATE, "rz_vision_forward: no position table for this patch grid (rz_set_position_table)");
    const int nv = it->second.n_valid, np = m->pad_tokens(nv, B);           // <= the table's rows (its padding is the batch-independent upper bound)
    if (np > it->second.n_pad) return fail(RZ_ERR_STATE, "rz_vision_forward: position table smaller than this batch's row padding (rz_set_position_table again after changing pad_rows)");
    if (B > m->cap_batch || np > m->cap_npad || (size_t)B * np > (size_t)m->cap_batch * m->cap_npad)
        return fail(RZ_ERR_STATE, "rz_vision_forward: workspace too small (rz_reserve)");
    const int D = m->D, H = m->H, F = m->F;
    const float eps = m->cfg.vit_layer_norm_eps;
    // Images are independent on this path.  The batch can be split into chunks, each with its own slice of every
    // intermediate buffer; with vision_streams = 2 the two halves run on two internal streams (fork/join by events on
    // the caller's stream) so that one half's MFMA-bound attention overlaps the other half's memory-bound GEMMs / LNs.
    auto run_chunk = [&](int c0, int Bc, hipStream_t s) -> int {
        int rc = 0;
        const int M = Bc * np;
        const size_t es = dsize(m->dt);
        const size_t row0 = (size_t)c0 * np;                                   // first token row of this chunk
        float* h = (float*)m->h.p + row0 * D;
        const float* pxc = px + (size_t)c0 * C * Himg * Wimg;
        // fp32 mode on the f16 matrix pipe: activations between the kernels travel as hi/lo f16 planes (6 bytes per element where they feed
        // a GEMM: [hi | lo | hi] along K; 4 where they feed the attention: hi plane, lo plane)
        const bool sp = m->dt == RZ_F32 && m->o_gemm_f32_split() && m->o_attn_f32_split() && !m->split_w.empty();
        // MX form of the split GEMMs (rz_common.h): every GEMM of the chunk or none (the producers write ONE operand form).  The 256 x 256
        // kernel that runs it needs M % 256 == 0.  Round 6: options 1 (default) and 2 take it wherever it applies — rounds 4-5 started it at 64 row tiles
        // ("where that kernel's grid fills the chip"; kept as option 3), but the step says otherwise: 1024^2 x 1 / 2 / 3 images +11 / +19 / +22 %,
        // 518^2 x 4 / 8 +13 / +27 %, 224^2 x 32 +30 % (profiles/r06/fp32_operand_form_small_batches_step_ab.txt), and one arithmetic for (nearly) every batch
        const int mxo = m->o_gemm_f32_mx();
        const bool mxg = sp && mxo != 0 && m->mx_weights_ok && M % 256 == 0 && (mxo != 3 || M >= 256 * 64);
        if (!m->force_exact) m->last_f32_form = mxg ? 2 : sp ? 1 : 0;
        const int mxa = mxg ? m->o_attn_f32_mx() : 0;                         // the attention's MX form rides on the GEMMs' (its ctx leaves in the MX form)
        // which of the attention's correction terms are dropped (flash_attn_split_kernel ABL): none by default; P V on the hi planes alone only with
        // attn_f32_pv = 1 ('f32_precision fast': 1.2e-3 on the outlier checkpoint G8, outside the 1e-3 contract)
        const int drop = m->o_f32_drop();
        const int attn_abl = (m->o_attn_f32_pv() != 0 && !(drop & 128) ? 1 : 0) | ((drop & 128) ? 2 : 0) | ((drop & 64) ? 4 : 0);
        const int mx = mxg ? (1 | (mxa >= 1 && !(attn_abl & 1) ? 2 : 0) | (mxa >= 2 ? 4 : 0)) : 0;       // V^T needs no pair plane when P V runs on the hi planes alone
        const size_t ex = m->dt == RZ_F32 ? 6 : es;
        char* xn = (char*)m->xn.p + row0 * D * ex;
        char* qkb = (char*)m->qk.p + row0 * 2 * D * es;
        char* vtb = (char*)m->vt.p + row0 * D * es;
        char* ctxb = (char*)m->ctx.p + row0 * D * ex;
        char* mid = (char*)m->mid.p + row0 * std::max((size_t)F, (size_t)m->KPAD) * ex;
        // auto: ~126 MiB of hidden activations per pass (4 images of 5376 rows at 16 bits), whole images only
        const size_t hid_row_bytes = (size_t)F * es;
        const int mlp_auto = (int)std::max<size_t>(1, ((size_t)132 << 20) / (hid_row_bytes * np));
        const int mlp_chunk = m->o_mlp_chunk();
        const int mlp_images = mlp_chunk > 0 ? mlp_chunk : (mlp_chunk == 0 ? Bc : std::min(Bc, mlp_auto));

        {   // patch embedding: im2col + GEMM with (pos | cls | bias) table epilogue
            ProfScope ps(m, RZ_PROF_ROWOPS, s);
            RZ_HIP(launch_im2col(m->dt, pxc, mid, Bc, C, Himg, Wimg, P, gh, gw, np, m->KPAD, s, m->run_if));
        }
        const int nblocks = (int)m->blocks.size();
        // LayerNorm fused into the GEMMs either side of it (gemm8.hip): every block of this chunk or none
        const bool fused = m->o_ln_fused() && nblocks > 0 && gemm_ln_fused_ok(m->dt, M, D, F, m->o_gemm_variant());
        float* part = fused ? (float*)m->lnpart.p + row0 * 24 : nullptr;
        float* stat = fused ? (float*)m->lnstat.p + row0 * 2 : nullptr;
        float* lnmu = fused ? (float*)m->lnmu.p + row0 : nullptr;
        // block 0's LayerNorm inputs (T copy of the embeddings times its gain, row statistics) straight from the patch GEMM's epilogue where the
        // persistent kernel runs it; otherwise the ln_prepare row kernel below
        bool patch_ln = false;
        if (fused && m->cfg.vit_layers > 0) {
            GemmArgs g;
            g.A = mid; g.lda = m->KPAD; g.W = m->patch_w.p; g.ldw = m->KPAD; g.M = M; g.N = D; g.K = m->KPAD; g.bias = nullptr; g.out = h; g.ldo = D;
            g.scale = (const float*)it->second.buf.p; g.resid = nullptr; g.ldr = 0; g.rows_per_image = np; g.heads_total = 0;
            g.ln_part = part; g.ln_hb = xn; g.ln_gamma = (const float*)m->blocks[0].ln1_g.p;
            g.variant = m->o_gemm_variant(); g.small_tile = m->o_gemm_small_tile(); g.raster = m->o_gemm_raster();
            if (gemm_patch_ln_ok(m->dt, g)) {
                patch_ln = true;
                {
                    ProfScope ps(m, RZ_PROF_GEMM, s);
                    RZ_HIP(launch_gemm(m->dt, EPI_PATCH_LN, g, s));
                }
                ProfScope ps(m, RZ_PROF_ROWOPS, s);
                // the epilogue centred with 0: told to the merge kernel, NOT a hipMemsetAsync of lnmu — inside a captured graph (torch.cuda.graph) a memset
                // placed between the kernels that write and read lnmu was not ordered with them on replay (first replay right, later ones wrong:
                // tests/test_gpu_fullsize.py::test_graph_replay_equals_eager_at_full_size); a plain capture of memset -> kernel replays correctly
                RZ_HIP(launch_ln_finalize(part, lnmu, stat, eps, M, s, true));
            }
        }
        if (!patch_ln && (rc = gemm(m, EPI_PATCH, mid, m->KPAD, m->patch_w.p, m->KPAD, M, D, m->KPAD, nullptr, h, D,
                                    (const float*)it->second.buf.p, nullptr, 0, np, 0, s, A_F32, false, 0, mx))) return rc;
        if (fused) {
            for (auto& b : m->blocks)
                if ((rc = fold_block(m, b))) return rc;
        }
        if (m->cfg.vit_layers == 0) {
            // no ViT blocks: Dinov2Model.layernorm (TF:dinov2/modeling_dinov2.py:469) acts on the embeddings, align blocks follow
            ProfScope ps(m, RZ_PROF_ROWOPS, s);
            if (fused) RZ_HIP(launch_ln_prepare(m->dt, h, (const float*)m->vit_ln_g.p, (const float*)m->vit_ln_b.p, eps, h, (const float*)m->blocks[0].ln1_g.p, xn, lnmu, stat, eps, M, D, s));
            else RZ_HIP(launch_layernorm(m->dt, h, (const float*)m->vit_ln_g.p, (const float*)m->vit_ln_b.p, eps, nullptr, h, M, D, s, m->run_if));
        } else if (fused && !patch_ln) {      // block 0 reads the patch-embedding output: copy + statistics by the row kernel
            ProfScope ps(m, RZ_PROF_ROWOPS, s);
            RZ_HIP(launch_ln_prepare(m->dt, h, nullptr, nullptr, 0.f, nullptr, (const float*)m->blocks[0].ln1_g.p, xn, lnmu, stat, eps, M, D, s));
        }
        for (int li = 0; li < nblocks; ++li) {
            const DinoBlock& b = m->blocks[li];
            const bool last_vit = (li == m->cfg.vit_layers - 1), last = (li == nblocks - 1);
            if (sp) {
                {
                    ProfScope ps(m, RZ_PROF_ROWOPS, s);
                    RZ_HIP(launch_layernorm_split3(h, (const float*)b.ln1_g.p, (const float*)b.ln1_b.p, eps, xn, M, D, (unsigned*)m->ovf.p, s, mx != 0));
                }
                // q | k -> hi / lo planes of [Bc][2H][np][64]; V^T -> hi / lo planes of [Bc][H][64][np]
                const char* wv = (const char*)b.wqkv.p + (size_t)2 * D * D * 4;
                bool paired = false;
                if ((rc = gemm_f32_split_qkv_pair(m, xn, b, M, np, qkb, vtb, mx, s, &paired))) return rc;
                if (!paired) {
                    if ((rc = gemm(m, EPI_HEADS, xn, D, b.wqkv.p, D, M, 2 * D, D, (const float*)b.bqkv.p, qkb, 0, nullptr, nullptr, 0, np, 2 * H, s, A_SPLIT, true,
                                   (int64_t)M * 2 * D, mx))) return rc;
                    if ((rc = gemm(m, EPI_VT, xn, D, wv, D, M, D, D, (const float*)b.bqkv.p + 2 * D, vtb, 0, nullptr, nullptr, 0, np, H, s, A_SPLIT, true, (int64_t)M * D, mx))) return rc;
                }
            } else if (!fused) {
                {
                    ProfScope ps(m, RZ_PROF_ROWOPS, s);
                    RZ_HIP(launch_layernorm(m->dt, h, (const float*)b.ln1_g.p, (const float*)b.ln1_b.p, eps, xn, nullptr, M, D, s, m->run_if));
                }
                // q | k | v projection (TF:dinov2/modeling_dinov2.py:199-213): ONE launch over N = 3D where the persistent kernel
                // applies (q, k -> per-head rows, v -> transposed, chosen per 256-column tile), else q|k and v separately
                if ((rc = gemm_qkv(m, xn, b, M, np, qkb, vtb, s))) return rc;
            } else {
                GemmArgs probe;
                probe.A = xn; probe.lda = D; probe.W = b.wqkv.p; probe.ldw = D; probe.M = M; probe.N = 3 * D; probe.K = D; probe.out2 = vtb; probe.split_n = 2 * D; probe.variant = m->o_gemm_variant(); probe.small_tile = m->o_gemm_small_tile(); probe.raster = m->o_gemm_raster();
                bool paired_first = false;
                if (m->o_gemm_qkv_pair_first(M) && (rc = gemm_ln_qkv_pair(m, xn, b, stat, M, np, qkb, vtb, s, &paired_first))) return rc;
                if (paired_first) {
                } else if (gemm_qkv_fused_ok(m->dt, probe)) {
                    if ((rc = gemm_ln(m, EPI_QKV_LN, xn, b.wqkv, b.c1qkv, b.c2qkv, stat, M, 3 * D, np, qkb, 0, 2 * H, vtb, H, 2 * D, s))) return rc;
                } else {        // small batches: the same projection as q|k and v launches of the 128x128 kernel
                    Tensor wv = b.wqkv, c1v = b.c1qkv, c2v = b.c2qkv;
                    wv.p = (char*)b.wqkv.p + (size_t)2 * D * D * es;
                    c1v.p = (float*)b.c1qkv.p + 2 * D;
                    c2v.p = (float*)b.c2qkv.p + 2 * D;
                    bool paired = false;
                    if ((rc = gemm_ln_qkv_pair(m, xn, b, stat, M, np, qkb, vtb, s, &paired))) return rc;
                    if (!paired) {
                        if ((rc = gemm_ln(m, EPI_HEADS_LN, xn, b.wqkv, b.c1qkv, b.c2qkv, stat, M, 2 * D, np, qkb, 0, 2 * H, nullptr, 0, 0, s))) return rc;
                        if ((rc = gemm_ln(m, EPI_VT_LN, xn, wv, c1v, c2v, stat, M, D, np, vtb, 0, H, nullptr, 0, 0, s))) return rc;
                    }
                }
            }
            {
                ProfScope ps(m, RZ_PROF_ATTN, s);
                // q heads are heads [0,H) and k heads [H,2H) of the [B][2H][np][64] tensor
                const char* qb = (const char*)qkb;
                const char* kb = qb + (size_t)H * np * 64 * es;
                if (sp)                                         // planes in (f16: k heads start H*np*64 ELEMENTS behind q), [hi | lo | hi] ctx out
                    RZ_HIP(launch_flash_attn_split_planes(qb, qb + (size_t)H * np * 64 * 2, vtb, ctxb, (int64_t)2 * H * np * 64, (int64_t)M * 2 * D, (int64_t)M * D,
                                                          Bc, H, nv, np, (unsigned*)m->ovf.p, s, mx != 0, mxa, attn_abl));
                else if (m->dt == RZ_F32 && m->o_attn_f32_split())      // hi/lo f16 planes of q, k, V^T live in `mid` (free between the QKV and fc1 GEMMs: 3/4 of it)
                    RZ_HIP(launch_flash_attn_f32_split((const float*)qb, (const float*)kb, (const float*)vtb, (float*)ctxb, mid, (int64_t)2 * H * np * 64,
                                                       Bc, H, nv, np, (unsigned*)m->ovf.p, s, 0, (attn_abl & 1) != 0));
                else
                    RZ_HIP(flash_attn(m->o_attn_variant(), m->dt, qb, kb, vtb, ctxb, (int64_t)2 * H * np * 64, Bc, H, nv, np, s, m->run_if));
            }
            if (sp) {
                if ((rc = gemm(m, EPI_RESID_SCALE, ctxb, D, b.wo.p, D, M, D, D, (const float*)b.bo.p, nullptr, 0, (const float*)b.ls1.p, h, D, np, 0, s, A_SPLIT, false, 0, mx))) return rc;
                {
                    ProfScope ps(m, RZ_PROF_ROWOPS, s);
                    RZ_HIP(launch_layernorm_split3(h, (const float*)b.ln2_g.p, (const float*)b.ln2_b.p, eps, xn, M, D, (unsigned*)m->ovf.p, s, mx != 0));
                }
            } else if (!fused) {
                if ((rc = gemm(m, EPI_RESID_SCALE, ctxb, D, b.wo.p, D, M, D, D, (const float*)b.bo.p, nullptr, 0, (const float*)b.ls1.p, h, D, np, 0, s))) return rc;
                {
                    ProfScope ps(m, RZ_PROF_ROWOPS, s);
                    RZ_HIP(launch_layernorm(m->dt, h, (const float*)b.ln2_g.p, (const float*)b.ln2_b.p, eps, xn, nullptr, M, D, s, m->run_if));
                }
            } else {
                if ((rc = gemm_resid_ln(m, ctxb, D, b.wo, b.bo, b.ls1, M, D, h, np, b.ln2_g, xn, part, lnmu, stat, eps, s))) return rc;
            }
            // MLP in row chunks that all reuse the FIRST rows of `mid`: the GELU'd hidden activations of a few images
            // (4 x 5376 x 3072 bf16 = 126 MiB) stay in the 256 MB Infinity Cache between fc1's stores and fc2's loads and
            // the same lines are overwritten by the next chunk, instead of 1 GB per layer going out to HBM and back.
            // Measured: tools/kmlp.py 2.05 -> 1.83 ms per layer of 32 images in isolation, but no gain inside the model
            // (372-374 vs 375-376 images/s), so the default is one pass over the whole chunk (option "mlp_chunk").
            if (fused) {
                if ((rc = gemm_ln(m, EPI_GELU_LN, xn, b.w1, b.c1_1, b.c2_1, stat, M, F, np, mid, F, 0, nullptr, 0, 0, s))) return rc;
                if (last || last_vit) {     // nobody reads this block's output through a fused LayerNorm: plain residual epilogue
                    if ((rc = gemm(m, EPI_RESID_SCALE, mid, F, b.w2.p, F, M, D, F, (const float*)b.b2.p, nullptr, 0, (const float*)b.ls2.p, h, D, np, 0, s))) return rc;
                } else {
                    if ((rc = gemm_resid_ln(m, mid, F, b.w2, b.b2, b.ls2, M, F, h, np, m->blocks[li + 1].ln1_g, xn, part, lnmu, stat, eps, s))) return rc;
                }
            } else if (sp) {        // fc1 writes [hi | lo | hi] of GELU(.) straight into fc2's A operand
                if ((rc = gemm(m, EPI_GELU, xn, D, b.w1.p, D, M, F, D, (const float*)b.b1.p, mid, F, nullptr, nullptr, 0, np, 0, s, A_SPLIT, true, 0, mx))) return rc;
                if ((rc = gemm(m, EPI_RESID_SCALE, mid, F, b.w2.p, F, M, D, F, (const float*)b.b2.p, nullptr, 0, (const float*)b.ls2.p, h, D, np, 0, s, A_SPLIT, false, 0, mx))) return rc;
            } else {
                for (int i0 = 0; i0 < Bc; i0 += mlp_images) {
                    const int Mi = std::min(mlp_images, Bc - i0) * np;
                    const size_t r0 = (size_t)i0 * np;
                    if ((rc = gemm(m, EPI_GELU, xn + r0 * D * es, D, b.w1.p, D, Mi, F, D, (const float*)b.b1.p, mid, F, nullptr, nullptr, 0, np, 0, s))) return rc;
                    if ((rc = gemm(m, EPI_RESID_SCALE, mid, F, b.w2.p, F, Mi, D, F, (const float*)b.b2.p, nullptr, 0, (const float*)b.ls2.p, h + r0 * D, D, np, 0, s))) return rc;
                }
            }
            if (last_vit) {   // Dinov2Model.layernorm (TF:dinov2/modeling_dinov2.py:469); align blocks follow
                ProfScope ps(m, RZ_PROF_ROWOPS, s);
                if (fused && !last)
                    RZ_HIP(launch_ln_prepare(m->dt, h, (const float*)m->vit_ln_g.p, (const float*)m->vit_ln_b.p, eps, h, (const float*)m->blocks[li + 1].ln1_g.p, xn, lnmu, stat, eps, M, D, s));
                else
                    RZ_HIP(launch_layernorm(m->dt, h, (const float*)m->vit_ln_g.p, (const float*)m->vit_ln_b.p, eps, nullptr, h, M, D, s, m->run_if));
            }
        }
        return 0;
    };
    const int nstreams = (m->o_vision_streams() == 2 && B >= 2) ? 2 : 1;
    if (nstreams == 2) {
        if (!m->side_ready) {
            for (int i = 0; i < 2; ++i) {
                RZ_HIP(hipStreamCreateWithFlags(&m->side[i], hipStreamNonBlocking));
                RZ_HIP(hipEventCreateWithFlags(&m->side_done[i], hipEventDisableTiming));
            }
            RZ_HIP(hipEventCreateWithFlags(&m->fork, hipEventDisableTiming));
            m->side_ready = true;
        }
        RZ_HIP(hipEventRecord(m->fork, s));
        const int half = (B + 1) / 2;
        for (int i = 0; i < 2; ++i) {
            RZ_HIP(hipStreamWaitEvent(m->side[i], m->fork, 0));
            const int c0 = i * half, Bc = i == 0 ? half : B - half;
            if ((rc = run_chunk(c0, Bc, m->side[i]))) return rc;
            RZ_HIP(hipEventRecord(m->side_done[i], m->side[i]));
            RZ_HIP(hipStreamWaitEvent(s, m->side_done[i], 0));
        }
    } else {
        const int vchunk = m->o_vision_chunk();
        const int chunk = (vchunk > 0 && vchunk < B) ? vchunk : B;
        for (int c0 = 0; c0 < B; c0 += chunk)
            if ((rc = run_chunk(c0, std::min(chunk, B - c0), s))) return rc;
    }
    m->last_batch = B;
    m->last_nvalid = nv;
    m->last_npad = np;
    if (tokens_out) {
        ProfScope ps(m, RZ_PROF_ROWOPS, s);
        RZ_HIP(launch_copy_tokens((const float*)m->h.p, tokens_out, B, nv, np, D, s, m->run_if));
    }
    return 0;
}

// planes: the fp32 mode's three-plane form — every GEMM operand travels as [hi | lo | hi] f16 planes written by its producer (one split pass for the
// embeddings, then the LayerNorms, the attention and fc1's epilogue write them): 7 launches per layer, as the 16-bit modes
static int text_forward_once(rz_handle_t m, const int64_t* ids, const int64_t* mask, int T, int L, const float* rel_bias, hipStream_t s, bool planes) {
    int rc = 0;
    const int D = m->D, H = m->H, TF = m->cfg.text_intermediate_size;
    const int rows = T * L, Mp = round_up(rows, 128);
    const float eps = m->cfg.text_layer_norm_eps;
    float* th = (float*)m->th.p;
    float* tsum = (float*)m->tsum.p;
    {
        ProfScope ps(m, RZ_PROF_ROWOPS, s);
        RZ_HIP(launch_text_embed(m->dt, ids, (const float*)m->word_emb.p, (const float*)m->pos_emb.p, (const float*)m->temb_ln_g.p,
                                 (const float*)m->temb_ln_b.p, eps, th, m->txn.p, T, L, D, m->cfg.vocab_size,
                                 m->cfg.max_position_embeddings, m->cfg.pad_token_id, s, m->run_if));
    }
    if (planes) {
        unsigned* flag = (unsigned*)m->ovf.p + 5;
        char* txn3 = (char*)m->tasplit.p;                                   // f16 planes: 2 bytes per element
        char* tctx3 = txn3 + (size_t)m->cap_trows * 3 * D * 2;
        char* tmid3 = tctx3 + (size_t)m->cap_trows * 3 * D * 2;
        {
            ProfScope ps(m, RZ_PROF_ROWOPS, s);
            RZ_HIP(launch_split3((const float*)m->txn.p, D, txn3, Mp, D, 0, flag, s));
        }
        auto tg = [&](int epi, const void* A, const Tensor& W, int N, int K, const Tensor& bias, void* out, int64_t ldo, float* resid, bool out_split) {
            GemmArgs g;
            g.A = A; g.lda = K; g.W = W.p; g.ldw = K; g.M = Mp; g.N = N; g.K = K; g.bias = (const float*)bias.p; g.out = out; g.ldo = ldo;
            g.scale = nullptr; g.resid = resid; g.ldr = D; g.rows_per_image = Mp; g.heads_total = 0; g.variant = m->o_gemm_variant(); g.small_tile = m->o_gemm_small_tile(); g.raster = m->o_gemm_raster();
            return gemm_text_split(m, epi, g, s, out_split);
        };
        for (const TextLayer& l : m->tlayers) {
            if ((rc = tg(EPI_STORE, txn3, l.wqkv, 3 * D, D, l.bqkv, m->tqkv.p, 3 * D, nullptr, false))) return rc;
            {
                ProfScope ps(m, RZ_PROF_ATTN, s);
                RZ_HIP(launch_text_attn(m->dt, m->tqkv.p, rel_bias, mask, m->tctx.p, T, L, H, s, nullptr, tctx3, flag));
            }
            if ((rc = tg(EPI_RESID_ADD, tctx3, l.wo, D, D, l.bo, tsum, D, th, false))) return rc;
            {
                ProfScope ps(m, RZ_PROF_ROWOPS, s);
                RZ_HIP(launch_layernorm_split3(tsum, (const float*)l.lna_g.p, (const float*)l.lna_b.p, eps, txn3, rows, D, flag, s, 0, th));
            }
            if ((rc = tg(EPI_GELU, txn3, l.w1, TF, D, l.b1, tmid3, TF, nullptr, true))) return rc;
            if ((rc = tg(EPI_RESID_ADD, tmid3, l.w2, D, TF, l.b2, tsum, D, th, false))) return rc;
            {
                ProfScope ps(m, RZ_PROF_ROWOPS, s);
                RZ_HIP(launch_layernorm_split3(tsum, (const float*)l.lno_g.p, (const float*)l.lno_b.p, eps, txn3, rows, D, flag, s, 0, th));
            }
        }
        return 0;
    }
    for (const TextLayer& l : m->tlayers) {
        if ((rc = gemm(m, EPI_STORE, m->txn.p, D, l.wqkv.p, D, Mp, 3 * D, D, (const float*)l.bqkv.p, m->tqkv.p, 3 * D, nullptr, nullptr, 0, Mp, 0, s))) return rc;
        {
            ProfScope ps(m, RZ_PROF_ATTN, s);
            RZ_HIP(launch_text_attn(m->dt, m->tqkv.p, rel_bias, mask, m->tctx.p, T, L, H, s, m->run_if));
        }
        if ((rc = gemm(m, EPI_RESID_ADD, m->tctx.p, D, l.wo.p, D, Mp, D, D, (const float*)l.bo.p, tsum, D, nullptr, th, D, Mp, 0, s))) return rc;
        {
            ProfScope ps(m, RZ_PROF_ROWOPS, s);
            RZ_HIP(launch_layernorm(m->dt, tsum, (const float*)l.lna_g.p, (const float*)l.lna_b.p, eps, m->txn.p, th, rows, D, s, m->run_if));
        }
        if ((rc = gemm(m, EPI_GELU, m->txn.p, D, l.w1.p, D, Mp, TF, D, (const float*)l.b1.p, m->tmid.p, TF, nullptr, nullptr, 0, Mp, 0, s))) return rc;
        if ((rc = gemm(m, EPI_RESID_ADD, m->tmid.p, TF, l.w2.p, TF, Mp, D, TF, (const float*)l.b2.p, tsum, D, nullptr, th, D, Mp, 0, s))) return rc;
        {
            ProfScope ps(m, RZ_PROF_ROWOPS, s);
            RZ_HIP(launch_layernorm(m->dt, tsum, (const float*)l.lno_g.p, (const float*)l.lno_b.p, eps, m->txn.p, th, rows, D, s, m->run_if));
        }
    }
    return 0;
}

int rz_text_forward(rz_handle_t m, const int64_t* ids, const int64_t* mask, int T, int L, const float* rel_bias, float* out, void* stream) {
    if (!m || !ids || !mask || !rel_bias || !out) return fail(RZ_ERR_INVALID, "rz_text_forward: null argument");
    if (T <= 0 || L <= 0) return fail(RZ_ERR_INVALID, "rz_text_forward: empty prompt batch");
    if (L + m->cfg.pad_token_id + 1 > m->cfg.max_position_embeddings)
        return fail(RZ_ERR_INVALID, "rz_text_forward: sequence longer than max_position_embeddings allows");
    int rc = rz_weights_ready(m);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int D = m->D;
    const int rows = T * L, Mp = round_up(rows, 128);
    if (Mp > m->cap_trows) return fail(RZ_ERR_STATE, "rz_text_forward: workspace too small (rz_reserve)");
    // fp32 mode (round 6): the GEMMs run on the three-plane f16 form (gemm_text_split); a value beyond the planes' range raises the text guard word and the
    // encode is repeated on the exact-fp32 kernels behind that word as a predicate — the vision forward's scheme (rz_vision_forward), on words of its own
    const bool split = m->dt == RZ_F32 && m->o_gemm_f32_split() && m->text_split_ok && m->tasplit.p && m->ovf.p;
    const bool guard = split && m->o_guard();
    if (split) RZ_HIP(launch_guard_word((unsigned*)m->ovf.p, 0, s, 5, 6));
    rc = text_forward_once(m, ids, mask, T, L, rel_bias, s, split);
    if (rc) return rc;
    if (guard) {
        m->force_exact = true;
        m->run_if = (const unsigned*)m->ovf.p + 5;
        rc = text_forward_once(m, ids, mask, T, L, rel_bias, s, false);
        m->force_exact = false;
        m->run_if = nullptr;
        if (rc) return rc;
        RZ_HIP(launch_guard_word((unsigned*)m->ovf.p, 1, s, 5, 6));
    }
    {
        ProfScope ps(m, RZ_PROF_ROWOPS, s);
        RZ_HIP(launch_masked_meanpool((const float*)m->th.p, mask, out, T, L, D, s));
    }
    return 0;
}

int rz_vlcabs(rz_handle_t m, const float* text_features, int T, int B, float* scores, float* t2i, float* logits, void* stream) {
    if (!m || !text_features || !scores || !t2i || !logits) return fail(RZ_ERR_INVALID, "rz_vlcabs: null argument");
    if (T <= 0 || B <= 0) return fail(RZ_ERR_INVALID, "rz_vlcabs: empty batch");
    if (B != m->last_batch) return fail(RZ_ERR_STATE, "rz_vlcabs: batch differs from the last rz_vision_forward");
    if (T > m->cap_prompts || B > m->cap_batch) return fail(RZ_ERR_STATE, "rz_vlcabs: workspace too small (rz_reserve)");
    if (!m->shared_ln_g.loaded || !m->tau_loaded) return fail(RZ_ERR_STATE, "rz_vlcabs: RadZeroLoss weights not loaded");
    hipStream_t s = (hipStream_t)stream;
    const int D = m->D;
    ProfScope ps(m, RZ_PROF_VLCABS, s);
    const bool dot = m->o_sim_dot();
    // losses.py:214-221: "cos" divides the cosines by the temperature (attn_temperature if the checkpoint has one, else the loss
    // temperature, :175-181), "dot" divides the raw products by sqrt(D); modeling.py:322-328 divides the logits by the loss temperature
    const float denom = dot ? sqrtf((float)D) : (m->attn_tau_loaded ? m->attn_tau : m->tau);
    RZ_HIP(launch_ln_l2norm(text_features, D, (const float*)m->shared_ln_g.p, (const float*)m->shared_ln_b.p,
                            m->cfg.shared_layer_norm_eps, (float*)m->qhat.p, T, D, dot ? 0 : 1, s));
    RZ_HIP(launch_vlcabs((const float*)m->h.p, (const float*)m->shared_ln_g.p, (const float*)m->shared_ln_b.p,
                         m->cfg.shared_layer_norm_eps, (const float*)m->qhat.p, denom, m->tau, dot ? 1 : 0, (float*)m->vws.p,
                         scores, t2i, logits, B, T, m->last_nvalid, m->last_npad, D, s));
    return 0;
}

int rz_upsample_maps(rz_handle_t m, const float* maps, int64_t map_stride, int n_maps, int grid, int out_h, int out_w,
                     int apply_sigmoid, float* out, void* stream) {
    return rz_upsample_maps_ex(m, maps, map_stride, n_maps, grid, out_h, out_w, apply_sigmoid, 0, out, stream);
}

int rz_upsample_maps_ex(rz_handle_t m, const float* maps, int64_t map_stride, int n_maps, int grid, int out_h, int out_w,
                        int apply_sigmoid, int keep_aspect_ratio, float* out, void* stream) {
    if (!maps || !out || n_maps <= 0 || grid <= 0 || out_h <= 0 || out_w <= 0 || map_stride < (int64_t)grid * grid)
        return fail(RZ_ERR_INVALID, "rz_upsample_maps: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (m) {
        ProfScope ps(m, RZ_PROF_POST, s);
        RZ_HIP(launch_upsample_bilinear(maps, map_stride, out, nullptr, n_maps, grid, out_h, out_w, apply_sigmoid, keep_aspect_ratio, s));
    } else {
        RZ_HIP(launch_upsample_bilinear(maps, map_stride, out, nullptr, n_maps, grid, out_h, out_w, apply_sigmoid, keep_aspect_ratio, s));
    }
    return 0;
}

int rz_grounding_points(rz_handle_t m, const float* maps, int64_t map_stride, int n_maps, int grid, int out_h, int out_w,
                        int32_t* xy_out, void* keys_ws, void* stream) {
    return rz_grounding_points_ex(m, maps, map_stride, n_maps, grid, out_h, out_w, 0, xy_out, keys_ws, stream);
}

int rz_grounding_points_ex(rz_handle_t m, const float* maps, int64_t map_stride, int n_maps, int grid, int out_h, int out_w,
                           int keep_aspect_ratio, int32_t* xy_out, void* keys_ws, void* stream) {
    if (!maps || !xy_out || !keys_ws || n_maps <= 0 || grid <= 0 || out_h <= 0 || out_w <= 0 || map_stride < (int64_t)grid * grid)
        return fail(RZ_ERR_INVALID, "rz_grounding_points: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (m) {
        ProfScope ps(m, RZ_PROF_POST, s);
        RZ_HIP(launch_grounding_points(maps, map_stride, (unsigned long long*)keys_ws, xy_out, n_maps, grid, out_h, out_w, keep_aspect_ratio, s));
        return 0;
    }
    RZ_HIP(launch_grounding_points(maps, map_stride, (unsigned long long*)keys_ws, xy_out, n_maps, grid, out_h, out_w, keep_aspect_ratio, s));
    return 0;
}

int rz_preprocess_image(const void* image, int src_dtype, int height, int width, int channels, int out_side, const int32_t* bounds_h,
                        const int32_t* coeffs_h, int ksize_h, const int32_t* bounds_v, const int32_t* coeffs_v, int ksize_v,
                        const float* mean3_host, const float* std3_host, float rescale, int minmax_normalize, void* workspace,
                        float* pixel_values_out, void* stream) {
    if (!image || !bounds_h || !coeffs_h || !bounds_v || !coeffs_v || !mean3_host || !std3_host || !workspace || !pixel_values_out)
        return fail(RZ_ERR_INVALID, "rz_preprocess_image: null argument");
    if (height <= 0 || width <= 0 || out_side <= 0 || (channels != 1 && channels != 3) || ksize_h <= 0 || ksize_v <= 0)
        return fail(RZ_ERR_INVALID, "rz_preprocess_image: bad shape");
    if (src_dtype < 0 || src_dtype > 2 || (!minmax_normalize && src_dtype != 0))
        return fail(RZ_ERR_INVALID, "rz_preprocess_image: source dtype (without min-max normalisation the image must be uint8)");
    unsigned* mm = (unsigned*)workspace;                            // first 16 bytes: min/max scratch
    unsigned char* ws8 = (unsigned char*)workspace + 16;
    RZ_HIP(launch_preprocess(image, src_dtype, height, width, channels, out_side, bounds_h, coeffs_h, ksize_h, bounds_v, coeffs_v, ksize_v,
                             mean3_host, std3_host, rescale, ws8, mm, pixel_values_out, minmax_normalize, (hipStream_t)stream));
    return 0;
}

namespace {
struct PreDescHost {       // = preprocess.hip's PreDesc
    const void* img; int dtype, H, W, C;
    int pad_left, pad_top, PH, PW;
    const int* bounds_h; const int* kk_h; int ksize_h;
    const int* bounds_v; const int* kk_v; int ksize_v;
    int64_t a8, b8, c8;
};
// lays the per-image byte buffers out behind the descriptor block; returns the total, 0 on a bad descriptor
size_t preprocess_layout(const rz_image_desc* d, int n, int S, std::vector<PreDescHost>* out, int* max_ph) {
    if (!d || n <= 0 || S <= 0) return 0;
    size_t off = preprocess_batch_desc_bytes(n);
    if (out) out->resize(n);
    int mph = 0;
    for (int i = 0; i < n; ++i) {
        const rz_image_desc& e = d[i];
        if (!e.image_dev || e.height <= 0 || e.width <= 0 || (e.channels != 1 && e.channels != 3) || e.src_dtype < 0 || e.src_dtype > 2) return 0;
        if (e.pad_left < 0 || e.pad_top < 0 || e.padded_height < e.height + e.pad_top || e.padded_width < e.width + e.pad_left) return 0;
        if (!e.bounds_h_dev || !e.coeffs_h_dev || !e.bounds_v_dev || !e.coeffs_v_dev || e.ksize_h <= 0 || e.ksize_v <= 0) return 0;
        const size_t C = e.channels, PH = e.padded_height, PW = e.padded_width;
        PreDescHost p{e.image_dev, e.src_dtype, e.height, e.width, e.channels, e.pad_left, e.pad_top, e.padded_height, e.padded_width,
                      e.bounds_h_dev, e.coeffs_h_dev, e.ksize_h, e.bounds_v_dev, e.coeffs_v_dev, e.ksize_v, 0, 0, 0};
        p.a8 = (int64_t)off; off += (PH * PW * C + 255) / 256 * 256;
        p.b8 = (int64_t)off; off += (PH * (size_t)S * C + 255) / 256 * 256;
        p.c8 = (int64_t)off; off += ((size_t)S * S * C + 255) / 256 * 256;
        if (out) (*out)[i] = p;
        mph = std::max(mph, e.padded_height);
    }
    if (max_ph) *max_ph = mph;
    return off;
}
}  // namespace

size_t rz_preprocess_batch_workspace(const rz_image_desc* descs, int n, int out_side) { return preprocess_layout(descs, n, out_side, nullptr, nullptr); }

int rz_preprocess_batch(const rz_image_desc* descs, int n, int out_side, const float* mean3_host, const float* std3_host, float rescale,
                        int minmax_normalize, void* workspace, size_t workspace_bytes, float* pixel_values_out, void* stream) {
    if (!descs || !mean3_host || !std3_host || !workspace || !pixel_values_out) return fail(RZ_ERR_INVALID, "rz_preprocess_batch: null argument");
    static_assert(sizeof(PreDescHost) % 8 == 0, "descriptor block keeps the min/max words aligned");
    std::vector<PreDescHost> dd;
    int max_ph = 0;
    const size_t need = preprocess_layout(descs, n, out_side, &dd, &max_ph);
    if (need == 0) return fail(RZ_ERR_INVALID, "rz_preprocess_batch: bad image descriptor (shape, dtype, padding or tables)");
    if (workspace_bytes < need) return fail(RZ_ERR_STATE, "rz_preprocess_batch: workspace too small (rz_preprocess_batch_workspace)");
    if (!minmax_normalize)
        for (int i = 0; i < n; ++i)
            if (descs[i].src_dtype != 0) return fail(RZ_ERR_INVALID, "rz_preprocess_batch: without min-max normalisation the images must be uint8");
    RZ_HIP(launch_preprocess_batch(dd.data(), n, max_ph, out_side, mean3_host, std3_host, rescale, (unsigned char*)workspace, pixel_values_out,
                                   minmax_normalize, (hipStream_t)stream));
    return 0;
}

int rz_gemm(int dtype, int epilogue, const void* a, const void* w, const float* bias, void* out, int M, int N, int K, void* stream) {
    if (!a || !w || !out) return fail(RZ_ERR_INVALID, "rz_gemm: null argument");
    if (epilogue != EPI_STORE && epilogue != EPI_GELU && epilogue != EPI_STORE_F32) return fail(RZ_ERR_INVALID, "rz_gemm: epilogue");
    if (M % 128 || N % 128) return fail(RZ_ERR_INVALID, "rz_gemm: M and N must be multiples of 128");
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = a; g.lda = K; g.W = w; g.ldw = K; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out = out; g.ldo = N;
    g.rows_per_image = M;
    g.variant = g_opt.gemm_variant; g.small_tile = g_opt.gemm_small_tile; g.raster = g_opt.gemm_raster;
    RZ_HIP(launch_gemm(dtype, epilogue, g, (hipStream_t)stream));
    return 0;
}

int rz_gemm_ex(int dtype, int epilogue, const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out,
               int64_t ldo, const float* scale, float* resid, int64_t ldr, int rows_per_image, int heads_total, int M, int N,
               int K, void* stream) {
    if (!a || !w) return fail(RZ_ERR_INVALID, "rz_gemm_ex: null argument");
    if (epilogue < 0 || epilogue > EPI_STORE_F32) return fail(RZ_ERR_INVALID, "rz_gemm_ex: epilogue");
    if (epilogue == EPI_RESID_SCALE ? (!scale || !resid) : !out) return fail(RZ_ERR_INVALID, "rz_gemm_ex: missing output / residual / scale");
    if ((epilogue == EPI_RESID_ADD && !resid) || (epilogue == EPI_PATCH && !scale)) return fail(RZ_ERR_INVALID, "rz_gemm_ex: missing residual / table");
    if (M % 128 || N % 128) return fail(RZ_ERR_INVALID, "rz_gemm_ex: M and N must be multiples of 128");
    if ((epilogue == EPI_HEADS || epilogue == EPI_VT || epilogue == EPI_PATCH) && (rows_per_image <= 0 || rows_per_image % 128 || M % rows_per_image))
        return fail(RZ_ERR_INVALID, "rz_gemm_ex: rows_per_image must be a multiple of 128 dividing M");
    GemmArgs g;
    g.A = a; g.lda = lda; g.W = w; g.ldw = ldw; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out = out; g.ldo = ldo;
    g.scale = scale; g.resid = resid; g.ldr = ldr; g.rows_per_image = rows_per_image > 0 ? rows_per_image : M; g.heads_total = heads_total;
    g.variant = g_opt.gemm_variant; g.small_tile = g_opt.gemm_small_tile; g.raster = g_opt.gemm_raster;
    RZ_HIP(launch_gemm(dtype, epilogue, g, (hipStream_t)stream));
    return 0;
}

// fp32 GEMM on the f16 / fp8 matrix pipes, kernel level (the forms the fp32 mode's vision encoder uses): out[m][n] += a[m][:] . w[n][:] + bias[n]
// (read-modify-write of the fp32 `out`, i.e. EPI_RESID_SCALE with LayerScale 1).  form 0: three f16 planes per operand ([hi | lo | hi] x
// [hi | hi | lo]); form 1: the MX form (f16 hi plane + block-scaled e4m3 correction planes, rz_common.h).  ws_a / ws_w: 6 (form 0) or 4
// (form 1) bytes per element of a / w; ones: N floats of 1.0 (the LayerScale vector).
int rz_gemm_f32_split(int form, const float* a, const float* w, const float* bias, const float* ones, float* out, void* ws_a, void* ws_w,
                      int M, int N, int K, void* stream) {
    if (!a || !w || !out || !ws_a || !ws_w || !ones || (form != 0 && form != 1)) return fail(RZ_ERR_INVALID, "rz_gemm_f32_split: bad argument");
    if (M % 256 || N % 256 || K % 64 || K < 128) return fail(RZ_ERR_INVALID, "rz_gemm_f32_split: M, N multiples of 256, K a multiple of 64 >= 128");
    hipStream_t s = (hipStream_t)stream;
    RZ_HIP(launch_split3(a, K, ws_a, M, K, form ? 2 : 0, nullptr, s));
    RZ_HIP(launch_split3(w, K, ws_w, N, K, form ? 3 : 1, nullptr, s));
    GemmArgs g;
    g.A = ws_a; g.W = ws_w; g.M = M; g.N = N; g.bias = bias; g.out = nullptr; g.ldo = 0; g.scale = ones; g.resid = out; g.ldr = N;
    g.rows_per_image = M; g.heads_total = 0;
    if (form) {
        g.lda = g.ldw = 2 * (int64_t)K; g.K = 2 * K;
        RZ_HIP(launch_gemm_v7_mx(EPI_RESID_SCALE, g, 0, s));
    } else {
        g.lda = g.ldw = 3 * (int64_t)K; g.K = 3 * K; g.variant = 7;
        RZ_HIP(launch_gemm_split_f32out(EPI_RESID_SCALE, g, s, false));
    }
    return 0;
}

int rz_gemm_qkv(int dtype, const void* x, const void* w, const float* bias, void* qk, void* vt, int rows_per_image, int heads, int M,
                int* fused_out, void* stream) {
    if (!x || !w || !qk || !vt) return fail(RZ_ERR_INVALID, "rz_gemm_qkv: null argument");
    if (dtype < 0 || dtype > 2 || heads <= 0 || M <= 0 || M % 128 || rows_per_image <= 0 || rows_per_image % 128 || M % rows_per_image)
        return fail(RZ_ERR_INVALID, "rz_gemm_qkv: M and rows_per_image must be multiples of 128, rows_per_image dividing M");
    const int D = heads * 64;
    GemmArgs g;
    g.A = x; g.lda = D; g.W = w; g.ldw = D; g.M = M; g.N = 3 * D; g.K = D; g.bias = bias; g.out = qk; g.ldo = 0;
    g.scale = nullptr; g.resid = nullptr; g.ldr = 0; g.rows_per_image = rows_per_image; g.heads_total = 2 * heads;
    g.out2 = vt; g.heads_total2 = heads; g.split_n = 2 * D; g.variant = g_opt.gemm_variant; g.small_tile = g_opt.gemm_small_tile; g.raster = g_opt.gemm_raster;
    const bool fused = gemm_qkv_fused_ok(dtype, g);
    if (fused_out) *fused_out = fused ? 1 : 0;
    if (fused) {
        RZ_HIP(launch_gemm(dtype, EPI_QKV, g, (hipStream_t)stream));
        return 0;
    }
    GemmArgs a = g;
    a.N = 2 * D; a.out2 = nullptr; a.split_n = 0;
    RZ_HIP(launch_gemm(dtype, EPI_HEADS, a, (hipStream_t)stream));
    a.N = D; a.W = (const char*)w + (size_t)2 * D * D * dsize(dtype); a.bias = bias ? bias + 2 * D : nullptr; a.out = vt; a.heads_total = heads;
    RZ_HIP(launch_gemm(dtype, EPI_VT, a, (hipStream_t)stream));
    return 0;
}

int rz_layernorm(int dtype, const float* in, const float* gamma, const float* beta, float eps, void* out_t, float* out_f32,
                 int64_t rows, int dim, void* stream) {
    if (!in || !gamma || !beta || (!out_t && !out_f32)) return fail(RZ_ERR_INVALID, "rz_layernorm: null argument");
    if (dim != 768) return fail(RZ_ERR_UNSUPPORTED, "rz_layernorm: dim must be 768");
    RZ_HIP(launch_layernorm(dtype, in, gamma, beta, eps, out_t, out_f32, rows, dim, (hipStream_t)stream));
    return 0;
}

int rz_flash_attention(int dtype, const void* q, const void* k, const void* vt, void* ctx, int B, int H, int n_valid, int n_pad,
                       void* stream) {
    if (!q || !k || !vt || !ctx) return fail(RZ_ERR_INVALID, "rz_flash_attention: null argument");
    if (n_pad % 128 || n_valid <= 0 || n_valid > n_pad) return fail(RZ_ERR_INVALID, "rz_flash_attention: n_pad must be a multiple of 128 >= n_valid > 0");
    RZ_HIP(flash_attn(g_opt.attn_variant, dtype, q, k, vt, ctx, (int64_t)H * n_pad * 64, B, H, n_valid, n_pad, (hipStream_t)stream));
    return 0;
}

size_t rz_flash_attention_split_workspace(int B, int H, int n_pad) { return flash_attn_split_workspace_bytes(B, H, n_pad); }

int rz_flash_attention_f32_split(const float* q, const float* k, const float* vt, float* ctx, void* ws, int B, int H, int n_valid, int n_pad,
                                 void* stream) {
    if (!q || !k || !vt || !ctx || !ws) return fail(RZ_ERR_INVALID, "rz_flash_attention_f32_split: null argument");
    if (n_pad % 128 || n_valid <= 0 || n_valid > n_pad) return fail(RZ_ERR_INVALID, "rz_flash_attention_f32_split: n_pad must be a multiple of 128 >= n_valid > 0");
    RZ_HIP(launch_flash_attn_f32_split(q, k, vt, ctx, ws, (int64_t)H * n_pad * 64, B, H, n_valid, n_pad, nullptr, (hipStream_t)stream, 0, g_opt.attn_f32_pv != 0));
    return 0;
}

int rz_flash_attention_f32_mx(const float* q, const float* k, const float* vt, float* ctx, void* ws, int B, int H, int n_valid, int n_pad,
                              void* stream) {
    if (!q || !k || !vt || !ctx || !ws) return fail(RZ_ERR_INVALID, "rz_flash_attention_f32_mx: null argument");
    if (n_pad % 128 || n_valid <= 0 || n_valid > n_pad) return fail(RZ_ERR_INVALID, "rz_flash_attention_f32_mx: n_pad must be a multiple of 128 >= n_valid > 0");
    RZ_HIP(launch_flash_attn_f32_split(q, k, vt, ctx, ws, (int64_t)H * n_pad * 64, B, H, n_valid, n_pad, nullptr, (hipStream_t)stream, g_opt.attn_f32_mx >= 2 ? 2 : 1, g_opt.attn_f32_pv != 0));
    return 0;
}

int rz_text_embed_ln(int dtype, const int64_t* ids, const float* word_emb, const float* pos_emb, const float* gamma, const float* beta, float eps,
                     float* h_out, void* xn_out, int T, int L, int vocab, int max_pos, int pad_id, void* stream) {
    if (!ids || !word_emb || !pos_emb || !gamma || !beta || !h_out || !xn_out) return fail(RZ_ERR_INVALID, "rz_text_embed_ln: null argument");
    if (dtype < 0 || dtype > 2 || T <= 0 || L <= 0 || vocab <= 0 || max_pos <= 0) return fail(RZ_ERR_INVALID, "rz_text_embed_ln: bad shape / dtype");
    RZ_HIP(launch_text_embed(dtype, ids, word_emb, pos_emb, gamma, beta, eps, h_out, xn_out, T, L, 768, vocab, max_pos, pad_id, (hipStream_t)stream));
    return 0;
}

int rz_text_attention(int dtype, const void* qkv, const float* rel_bias, const int64_t* mask, void* ctx, int T, int L, int H, void* stream) {
    if (!qkv || !rel_bias || !mask || !ctx) return fail(RZ_ERR_INVALID, "rz_text_attention: null argument");
    if (dtype < 0 || dtype > 2 || T <= 0 || L <= 0 || H <= 0) return fail(RZ_ERR_INVALID, "rz_text_attention: bad shape / dtype");
    RZ_HIP(launch_text_attn(dtype, qkv, rel_bias, mask, ctx, T, L, H, (hipStream_t)stream));
    return 0;
}

int rz_masked_meanpool(const float* h, const int64_t* mask, float* out, int T, int L, int D, void* stream) {
    if (!h || !mask || !out || T <= 0 || L <= 0 || D <= 0) return fail(RZ_ERR_INVALID, "rz_masked_meanpool: bad argument");
    RZ_HIP(launch_masked_meanpool(h, mask, out, T, L, D, (hipStream_t)stream));
    return 0;
}

int rz_rows_dot(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, float* out, int M, int N, int K, int rows_per_group,
                int64_t out_group_stride, int64_t out_row_stride, int64_t out_col_stride, void* stream) {
    if (!a || !b || !out || M <= 0 || N <= 0 || K <= 0 || K % 4 || lda % 4 || ldb % 4 || lda < K || ldb < K || rows_per_group <= 0)
        return fail(RZ_ERR_INVALID, "rz_rows_dot: bad argument (K and the leading dimensions must be multiples of 4)");
    RZ_HIP(launch_rows_dot(a, lda, b, ldb, bias, out, M, N, K, rows_per_group, out_group_stride, out_row_stride, out_col_stride, (hipStream_t)stream));
    return 0;
}

int rz_image_features(const float* tokens, int64_t image_stride_rows, int batch, int n_tokens, int dim, float* out, void* stream) {
    if (!tokens || !out || batch <= 0 || n_tokens < 2 || dim <= 0 || dim % 64 || image_stride_rows < n_tokens)
        return fail(RZ_ERR_INVALID, "rz_image_features: bad argument (dim must be a multiple of 64, n_tokens >= 2)");
    RZ_HIP(launch_image_features(tokens, image_stride_rows, batch, n_tokens, dim, out, (hipStream_t)stream));
    return 0;
}

int rz_patch_embed(int dtype, const float* px, int B, int C, int Himg, int Wimg, int P, const void* weight, int k_pad, const float* table,
                   int n_pad, void* ws, float* out, void* stream) {
    if (!px || !weight || !table || !ws || !out) return fail(RZ_ERR_INVALID, "rz_patch_embed: null argument");
    if (dtype < 0 || dtype > 2 || B <= 0 || C <= 0 || P <= 0) return fail(RZ_ERR_INVALID, "rz_patch_embed: bad shape / dtype");
    const int gh = Himg / P, gw = Wimg / P;
    if (gh <= 0 || gw <= 0 || n_pad % 128 || n_pad < 1 + gh * gw || k_pad % 64 || k_pad < C * P * P)
        return fail(RZ_ERR_INVALID, "rz_patch_embed: n_pad must be a multiple of 128 >= 1 + grid, k_pad a multiple of 64 >= channels*patch^2");
    hipStream_t s = (hipStream_t)stream;
    RZ_HIP(launch_im2col(dtype, px, ws, B, C, Himg, Wimg, P, gh, gw, n_pad, k_pad, s));
    GemmArgs g;
    g.A = ws; g.lda = k_pad; g.W = weight; g.ldw = k_pad; g.M = B * n_pad; g.N = 768; g.K = k_pad; g.bias = nullptr; g.out = out; g.ldo = 768;
    g.scale = table; g.resid = nullptr; g.ldr = 0; g.rows_per_image = n_pad; g.heads_total = 0; g.variant = g_opt.gemm_variant; g.small_tile = g_opt.gemm_small_tile; g.raster = g_opt.gemm_raster;
    RZ_HIP(launch_gemm(dtype, EPI_PATCH, g, s));
    return 0;
}

int rz_debug_buffer(const char* what, void* dev_ptr) {
    if (!what) return fail(RZ_ERR_INVALID, "rz_debug_buffer: null name");
#ifdef RZ_EXPERIMENTS
    if (!strcmp(what, "gemm_v8_stamps")) { gemm_v8_set_stamp_buffer(dev_ptr); return 0; }
#else
    (void)dev_ptr;
#endif
    return fail(RZ_ERR_INVALID, std::string("rz_debug_buffer: unknown buffer ") + what + " (diagnostic buffers exist only in the -DRZ_EXPERIMENTS tools build)");
}

int rz_set_option(const char* name, int value) {
    if (!name) return fail(RZ_ERR_INVALID, "rz_set_option: null name");
    if (!strcmp(name, "gemm_v1_only")) { g_opt.gemm_variant = value ? 1 : 0; return 0; }
    int* f = option_field(g_opt, name);
    if (!f) return fail(RZ_ERR_INVALID, std::string("rz_set_option: unknown option ") + name);
    if (f == &g_opt.f32_drop && !f32_drop_ok(value)) return fail(RZ_ERR_INVALID, "rz_set_option: f32_drop bits 64 / 128 (scores / P on the hi planes alone) exist only in the -DRZ_EXPERIMENTS tools build");
    if (f == &g_opt.gemm_small_tile && (value == RZ_OPT_INHERIT || !small_tile_ok(value))) return fail(RZ_ERR_INVALID, "rz_set_option: gemm_small_tile is geometry (0 - 3) + 10 x stages (0, 2 or 4)");
    *f = value;
    return 0;
}

int rz_set_model_option(rz_handle_t m, const char* name, int value) {
    if (!m || !name) return fail(RZ_ERR_INVALID, "rz_set_model_option: null argument");
    int* f = option_field(m->opt, name);
    if (!f) return fail(RZ_ERR_INVALID, std::string("rz_set_model_option: unknown option ") + name);
    if (f == &m->opt.f32_drop && !f32_drop_ok(value)) return fail(RZ_ERR_INVALID, "rz_set_model_option: f32_drop bits 64 / 128 (scores / P on the hi planes alone) exist only in the -DRZ_EXPERIMENTS tools build");
    if (f == &m->opt.gemm_small_tile && !small_tile_ok(value)) return fail(RZ_ERR_INVALID, "rz_set_model_option: gemm_small_tile is geometry (0 - 3) + 10 x stages (0, 2 or 4)");
    if (f == &m->opt.pad_rows) {
        if (value != RZ_OPT_INHERIT && value != 0 && value != 128 && value != 256) return fail(RZ_ERR_INVALID, "rz_set_model_option: pad_rows is 0, 128 or 256");
        RZ_HIP(hipDeviceSynchronize());
        m->pos_tables.clear();           // tables and workspaces were sized by the old rule: the caller sets them again
        m->cap_batch = m->cap_npad = 0;
        m->last_batch = 0;
    }
    *f = value;
    return 0;
}

int rz_get_model_option(rz_handle_t m, const char* name, int* value_out) {
    if (!m || !name || !value_out) return fail(RZ_ERR_INVALID, "rz_get_model_option: null argument");
    if (!strcmp(name, "f32_split_guard_reruns")) {
        // counted on the device behind each guarded forward: wait for the device's queued work, then read the word (not callable while a stream is capturing)
        unsigned n = 0;
        if (m->ovf.p && m->ovf.bytes >= 32) {
            RZ_HIP(hipDeviceSynchronize());
            unsigned w[3] = {0, 0, 0};
            RZ_HIP(hipMemcpy(w, (const unsigned*)m->ovf.p + 4, 12, hipMemcpyDeviceToHost));
            n = w[0] + w[2];                 // vision forwards + prompt encodes repeated on the exact kernels
        }
        *value_out = (int)std::min<unsigned>(n, (unsigned)INT32_MAX);
        return 0;
    }
    // read-only facts of the last rz_vision_forward (bench.py prices its roofline from these instead of re-deriving the padding rule)
    if (!strcmp(name, "last_npad")) { *value_out = m->last_npad; return 0; }
    if (!strcmp(name, "last_batch")) { *value_out = m->last_batch; return 0; }
    if (!strcmp(name, "last_f32_form")) { *value_out = m->last_f32_form; return 0; }
    Options own = m->opt, proc = g_opt;
    int* fo = option_field(own, name);
    int* fp = option_field(proc, name);
    if (!fo) return fail(RZ_ERR_INVALID, std::string("rz_get_model_option: unknown option ") + name);
    *value_out = pick(*fo, *fp);
    return 0;
}

int rz_profile_enable(rz_handle_t m, int enable) {
    if (!m) return fail(RZ_ERR_INVALID, "null handle");
    m->prof = enable != 0;
    m->prof_mask = enable > 1 ? ((unsigned)enable >> 1) & 0x1Fu : 0x1Fu;      // enable = 1: every family; 1 | mask << 1: only the families in mask
    m->ev_used = 0;
    for (int i = 0; i < RZ_PROF_NFAM; ++i) { m->prof_ms[i] = 0.f; m->prof_n[i] = 0; }
    return 0;
}

int rz_profile_read(rz_handle_t m, float* ms, int64_t* n) {
    if (!m || !ms || !n) return fail(RZ_ERR_INVALID, "rz_profile_read: null argument");
    for (size_t i = 0; i < m->ev_used; ++i) {
        float t = 0.f;
        RZ_HIP(hipEventElapsedTime(&t, m->ev_pool[i].a, m->ev_pool[i].b));
        m->prof_ms[m->ev_pool[i].fam] += t;
        m->prof_n[m->ev_pool[i].fam] += 1;
    }
    m->ev_used = 0;
    for (int i = 0; i < RZ_PROF_NFAM; ++i) { ms[i] = m->prof_ms[i]; n[i] = m->prof_n[i]; m->prof_ms[i] = 0.f; m->prof_n[i] = 0; }
    return 0;
}

}  // extern "C"
