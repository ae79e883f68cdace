// Host-side sanitizer build (tools/host_asan.sh): stand-ins for the kernel launchers of rz_kernels.h.  No kernel runs here; instead
// every stub asks AddressSanitizer whether the byte ranges the real kernel would READ or WRITE lie inside live allocations
// (__asan_region_is_poisoned), with the extents the kernels' own indexing implies (written from the kernel sources, file:line below).
// So a workspace that rz_reserve sized too small, a weight buffer packed with the wrong leading dimension or a stale pointer after a
// reload is reported on the CPU box with a stack trace, without a GPU.  Outputs are filled with a finite pattern so that host code
// reading results back (overflow-guard words, profile counters) sees defined data.
#include <algorithm>
#include <sanitizer/asan_interface.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rz_kernels.h"

namespace rz {

enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
static size_t esz(int dt) { return dt == DT_F32 ? 4 : 2; }

static int g_checked = 0;
static void rd(const void* p, size_t bytes, const char* what) {
    if (!bytes) return;
    if (!p) { fprintf(stderr, "[host_asan] NULL pointer for %s\n", what); abort(); }
    if (void* bad = __asan_region_is_poisoned(const_cast<void*>(p), bytes)) {
        fprintf(stderr, "[host_asan] %s: %zu bytes at %p leave their allocation at %p\n", what, bytes, p, bad);
        __asan_describe_address(bad);
        abort();
    }
    ++g_checked;
}
static void wr(void* p, size_t bytes, const char* what) {
    rd(p, bytes, what);
    memset(p, 0, bytes);           // a write ASAN itself sees; zeros are finite in every dtype
}
int host_asan_ranges_checked() { return g_checked; }

// ---- GEMM family (gemm.hip / gemm7.hip / gemm8.hip; epilogue extents: gemm_common.h, gemm8_epilogue.h) -------------------------
static void gemm_ranges(int dtype, int epi, const GemmArgs& g, size_t out_es) {
    const size_t es = esz(dtype);
    rd(g.A, ((size_t)(g.M - 1) * g.lda + g.K) * es, "gemm A");
    rd(g.W, ((size_t)(g.N - 1) * g.ldw + g.K) * es, "gemm W");
    const size_t images = g.rows_per_image > 0 ? (size_t)g.M / g.rows_per_image : 1;
    const bool ln_consumer = epi == EPI_QKV_LN || epi == EPI_GELU_LN || epi == EPI_HEADS_LN || epi == EPI_VT_LN;
    if (ln_consumer) {
        rd(g.ln_stat, (size_t)g.M * 2 * 4, "gemm ln_stat");
        rd(g.scale, (size_t)g.N * 4, "gemm c1");
        rd(g.bias, (size_t)g.N * 4, "gemm c2");
    } else if (g.bias) {
        rd(g.bias, (size_t)g.N * 4, "gemm bias");
    }
    switch (epi) {
        case EPI_STORE: case EPI_GELU: case EPI_GELU_LN:
            wr(g.out, ((size_t)(g.M - 1) * g.ldo + g.N) * out_es, "gemm out (row-major)"); break;
        case EPI_STORE_F32:
            wr(g.out, ((size_t)(g.M - 1) * g.ldo + g.N) * 4, "gemm out (fp32)"); break;
        case EPI_RESID_ADD:
            rd(g.resid, ((size_t)(g.M - 1) * g.ldr + g.N) * 4, "gemm resid");
            wr(g.out, ((size_t)(g.M - 1) * g.ldo + g.N) * 4, "gemm out (fp32)"); break;
        case EPI_PATCH: case EPI_PATCH_LN:
            rd(g.scale, (size_t)g.rows_per_image * g.N * 4, "gemm patch table");
            wr(g.out, ((size_t)(g.M - 1) * g.ldo + g.N) * 4, "gemm out (fp32)");
            if (epi == EPI_PATCH_LN) {
                rd(g.ln_gamma, (size_t)g.N * 4, "gemm ln_gamma");
                wr(g.ln_part, (size_t)g.M * 12 * 2 * 4, "gemm ln_part");
                wr(g.ln_hb, (size_t)g.M * g.N * es, "gemm ln_hb");
            }
            break;
        case EPI_RESID_SCALE: case EPI_RESID_SCALE_LN:
            rd(g.scale, (size_t)g.N * 4, "gemm LayerScale");
            wr(g.resid, ((size_t)(g.M - 1) * g.ldr + g.N) * 4, "gemm residual stream");
            if (epi == EPI_RESID_SCALE_LN) {
                rd(g.ln_gamma, (size_t)g.N * 4, "gemm ln_gamma");
                rd(g.ln_mu, (size_t)g.M * 4, "gemm ln_mu");
                wr(g.ln_part, (size_t)g.M * 12 * 2 * 4, "gemm ln_part");
                wr(g.ln_hb, (size_t)g.M * g.N * es, "gemm ln_hb");
            }
            break;
        case EPI_HEADS: case EPI_HEADS_LN: case EPI_VT: case EPI_VT_LN:      // [b][heads_total][tok][64] | [b][heads_total][64][tok]
            wr(g.out, images * g.heads_total * g.rows_per_image * 64 * out_es, "gemm per-head out"); break;
        case EPI_QKV: case EPI_QKV_LN:
            wr(g.out, images * g.heads_total * g.rows_per_image * 64 * out_es, "gemm q|k out");
            wr(g.out2, images * g.heads_total2 * g.rows_per_image * 64 * out_es, "gemm v^T out"); break;
        default: fprintf(stderr, "[host_asan] unknown epilogue %d\n", epi); abort();
    }
}

hipError_t launch_gemm(int dtype, int epi, const GemmArgs& g, hipStream_t) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.M % 128 || g.N % 128) return hipErrorInvalidValue;
    if (dtype == DT_F32 && g.run_if) { rd(g.run_if, 4, "gemm predicate"); if (*g.run_if == 0) return hipSuccess; }      // exact-fp32 kernels: predicated launch
    gemm_ranges(dtype, epi, g, esz(dtype));
    return hipSuccess;
}
bool gemm_pair_ok(int dtype, int epi_a, const GemmArgs& ga, int epi_b, const GemmArgs& gb) {
    if (dtype == DT_F32 || !((epi_a == EPI_HEADS_LN && epi_b == EPI_VT_LN) || (epi_a == EPI_HEADS && epi_b == EPI_VT))) return false;
    return ga.A == gb.A && ga.lda == gb.lda && ga.M == gb.M && ga.K == gb.K && ga.M > 0 && ga.M % 128 == 0 && ga.N % 128 == 0 && gb.N % 128 == 0;
}
hipError_t launch_gemm_pair(int dtype, int epi_a, const GemmArgs& ga, int epi_b, const GemmArgs& gb, hipStream_t s) {
    if (!gemm_pair_ok(dtype, epi_a, ga, epi_b, gb)) return hipErrorInvalidValue;
    hipError_t e = launch_gemm(dtype, epi_a, ga, s);
    return e != hipSuccess ? e : launch_gemm(dtype, epi_b, gb, s);
}
bool gemm_pair_f32_ok(int, const GemmArgs&, const GemmArgs&, int) { return false; }      // the host build runs the two launches (same accesses)
hipError_t launch_gemm_pair_f32(int, const GemmArgs&, const GemmArgs&, int, hipStream_t) { return hipErrorInvalidValue; }
// fp32 mode: f16 planes along K (g.K = 3 K), fp32 or hi/lo-split outputs (gemm.hip launch_gemm_split_f32out)
hipError_t launch_gemm_split_f32out(int epi, const GemmArgs& g, hipStream_t s, bool split_out) {
    if (epi == EPI_RESID_SCALE || epi == EPI_RESID_ADD || epi == EPI_PATCH || epi == EPI_STORE_F32) return launch_gemm(DT_F16, epi, g, s);
    rd(g.A, ((size_t)(g.M - 1) * g.lda + g.K) * 2, "split gemm A planes");
    rd(g.W, ((size_t)(g.N - 1) * g.ldw + g.K) * 2, "split gemm W planes");
    if (g.bias) rd(g.bias, (size_t)g.N * 4, "split gemm bias");
    const size_t images = g.rows_per_image > 0 ? (size_t)g.M / g.rows_per_image : 1;
    if (!split_out) {
        if (epi == EPI_HEADS || epi == EPI_VT) wr(g.out, images * g.heads_total * g.rows_per_image * 64 * 4, "split gemm per-head fp32 out");
        else wr(g.out, ((size_t)(g.M - 1) * g.ldo + g.N) * 4, "split gemm fp32 out");
    } else if (epi == EPI_GELU) {
        wr(g.out, ((size_t)(g.M - 1) * 3 * g.ldo + 2 * g.ldo + g.N) * 2, "split gemm [hi|lo|hi] out");       // gemm_common.h: rows of 3 * ldo f16
    } else {
        const size_t plane = images * g.heads_total * g.rows_per_image * 64;
        wr(g.out, plane * 2, "split gemm hi plane");
        wr((char*)g.out + (size_t)g.plane_off * 2, plane * 2, "split gemm lo plane");
    }
    if (g.ovf_flag) wr(g.ovf_flag, 0, "ovf");
    return hipSuccess;
}
hipError_t launch_absmax_bits(const float* src, int64_t n, unsigned* out_bits, hipStream_t) {
    unsigned best = 0;
    for (int64_t i = 0; i < n; ++i) { unsigned u; memcpy(&u, src + i, 4); u &= 0x7fffffffu; best = u > best ? u : best; }
    *out_bits = best > *out_bits ? best : *out_bits;
    return hipSuccess;
}
hipError_t launch_split3(const float* src, int64_t ld, void* dst, int64_t rows, int K, int layout, unsigned* ovf, hipStream_t, int) {
    rd(src, ((size_t)(rows - 1) * ld + K) * 4, "split3 src");
    wr(dst, layout >= 2 ? (size_t)rows * 4 * K : (size_t)rows * 3 * K * 2, "split3 dst");          // 2 / 3: the MX form, 4 K bytes per row
    if (ovf) rd(ovf, 4, "split3 overflow word");
    return hipSuccess;
}
// fp32 mode, MX form (gemm7.hip launch_gemm_v7_mx): rows of 4 K bytes, g.K = g.lda = g.ldw = 2 K f16-element units
bool gemm_v7_mx_ok(const GemmArgs& g) { return g.M % 256 == 0 && g.N % 256 == 0 && g.K % 128 == 0 && g.K >= 256; }
hipError_t launch_gemm_v7_mx(int epi, const GemmArgs& g, int out_kind, hipStream_t) {
    if (!gemm_v7_mx_ok(g)) return hipErrorInvalidValue;
    rd(g.A, ((size_t)(g.M - 1) * g.lda + g.K) * 2, "MX gemm A rows");
    rd(g.W, ((size_t)(g.N - 1) * g.ldw + g.K) * 2, "MX gemm W rows");
    if (g.bias) rd(g.bias, (size_t)g.N * 4, "MX gemm bias");
    const size_t images = g.rows_per_image > 0 ? (size_t)g.M / g.rows_per_image : 1;
    if (out_kind == 0) {
        if (epi == EPI_PATCH) { rd(g.scale, (size_t)g.rows_per_image * g.N * 4, "patch table"); wr(g.out, ((size_t)(g.M - 1) * g.ldo + g.N) * 4, "MX gemm fp32 out"); }
        else { rd(g.scale, (size_t)g.N * 4, "LayerScale"); wr(g.resid, ((size_t)(g.M - 1) * g.ldr + g.N) * 4, "MX gemm residual stream"); }
    } else if (out_kind == 1 || out_kind == 3) {
        const size_t plane = images * g.heads_total * g.rows_per_image * 64;
        wr(g.out, plane * 2, "MX gemm hi plane");
        wr((char*)g.out + (size_t)g.plane_off * 2, plane * 2, "MX gemm lo plane");
    } else {
        wr(g.out, (size_t)g.M * 4 * g.ldo, "MX gemm MX-form out");
    }
    return hipSuccess;
}
hipError_t launch_gemm_v8_mx(int epi, int out_kind, const GemmArgs& g, hipStream_t s) { return launch_gemm_v7_mx(epi, g, out_kind, s); }      // same operands, same outputs
// round 6: the same form on the 128 x 128 kernel (gemm.hip launch_gemm_small_mx): same operands, same outputs; taken for small shapes
bool gemm_small_mx_ok(int epi, int out_kind, const GemmArgs& g) {
    if (g.M <= 0 || g.M % 128 || g.N % 128 || g.K % 128 || g.K < 256) return false;
    return (out_kind == 0 && (epi == EPI_RESID_SCALE || epi == EPI_PATCH)) || (out_kind == 1 && (epi == EPI_HEADS || epi == EPI_VT)) || (out_kind == 2 && epi == EPI_GELU) || (out_kind == 3 && epi == EPI_VT);
}
bool gemm_small_mx_pays(int, const GemmArgs& g) { return (int64_t)(g.M / 256) * (g.N / 256) < 128; }
hipError_t launch_gemm_small_mx(int epi, const GemmArgs& g, int out_kind, hipStream_t s) { return gemm_small_mx_ok(epi, out_kind, g) && gemm_v7_mx_ok(g) ? launch_gemm_v7_mx(epi, g, out_kind, s) : hipErrorInvalidValue; }
static bool big_tiles_pay(const GemmArgs& g) {          // gemm.hip
    if (g.M % 256 || g.M < 1024 || g.N % 256) return false;
    const int64_t t = (int64_t)(g.M / 256) * (g.N / 256);
    return t >= 128 && (double)t / (double)(((t + 255) / 256) * 256) >= 0.55;
}
bool gemm_v8_ok(int dtype, int, const GemmArgs& g) { return dtype != DT_F32 && g.M % 256 == 0 && g.N % 256 == 0 && g.K % 128 == 0 && g.K >= 256; }
bool gemm_v8_mx_ok(int epi, int out_kind, const GemmArgs& g) {
    const bool form = (out_kind == 0 && (epi == EPI_RESID_SCALE || epi == EPI_PATCH)) || (out_kind == 1 && epi == EPI_HEADS) || (out_kind == 2 && epi == EPI_GELU) || (out_kind == 3 && epi == EPI_VT);
    return form && gemm_v8_ok(DT_F16, epi, g);
}
bool gemm_qkv_fused_ok(int dtype, const GemmArgs& g) { return (g.variant == 0 || g.variant == 8) && gemm_v8_ok(dtype, EPI_QKV, g) && (g.variant != 0 || big_tiles_pay(g)); }
bool gemm_patch_ln_ok(int dtype, const GemmArgs& g) {
    return dtype != DT_F32 && (g.variant == 0 || g.variant == 1 || g.variant == 8) && g.M > 0 && g.M % 128 == 0 && g.N == 768 && g.K % 64 == 0 && g.ln_part && g.ln_hb &&
           g.ln_gamma && g.scale && g.out && g.rows_per_image > 0;
}
bool gemm_ln_fused_ok(int dtype, int M, int D, int F, int variant) { return dtype != DT_F32 && (variant == 0 || variant == 1 || variant == 8) && M > 0 && M % 128 == 0 && D == 768 && F % 128 == 0; }

// ---- attention (attention.hip) -----------------------------------------------------------------------------------------------
hipError_t launch_flash_attn(int dtype, const void* q, const void* k, const void* vT, void* ctx, int64_t bs, int B, int H, int nv, int np, int, hipStream_t, const unsigned* run_if) {
    if (np % 128 || nv <= 0 || nv > np) return hipErrorInvalidValue;
    if (run_if) { rd(run_if, 4, "attention predicate"); if (*run_if == 0) return hipSuccess; }
    const size_t es = esz(dtype), head = (size_t)np * 64;
    rd(q, ((size_t)(B - 1) * bs + H * head) * es, "attention q");
    rd(k, ((size_t)(B - 1) * bs + H * head) * es, "attention k");
    rd(vT, (size_t)B * H * head * es, "attention v^T");
    wr(ctx, (size_t)B * np * H * 64 * es, "attention ctx");
    return hipSuccess;
}
size_t flash_attn_split_workspace_bytes(int B, int H, int n_pad) { return (size_t)3 * B * H * n_pad * 64 * 4; }      // attention.hip
hipError_t launch_flash_attn_f32_split(const float* q, const float* k, const float* vT, float* ctx, void* ws, int64_t bs, int B, int H, int nv, int np, unsigned*, hipStream_t, int, int) {
    const size_t head = (size_t)np * 64;
    rd(q, ((size_t)(B - 1) * bs + H * head) * 4, "split attention q");
    rd(k, ((size_t)(B - 1) * bs + H * head) * 4, "split attention k");
    rd(vT, (size_t)B * H * head * 4, "split attention v^T");
    wr(ws, flash_attn_split_workspace_bytes(B, H, np), "split attention planes");
    wr(ctx, (size_t)B * np * H * 64 * 4, "split attention ctx");
    return hipSuccess;
}
hipError_t launch_flash_attn_split_planes(const void* q_hi, const void* k_hi, const void* v_hi, void* ctx3, int64_t bs, int64_t qk_lo, int64_t v_lo, int B, int H, int nv, int np, unsigned*, hipStream_t, int mx_out, int, int) {
    const size_t head = (size_t)np * 64;
    for (int plane = 0; plane < 2; ++plane) {
        rd((const char*)q_hi + (size_t)plane * qk_lo * 2, ((size_t)(B - 1) * bs + H * head) * 2, "split attention q plane");
        rd((const char*)k_hi + (size_t)plane * qk_lo * 2, ((size_t)(B - 1) * bs + H * head) * 2, "split attention k plane");
        rd((const char*)v_hi + (size_t)plane * v_lo * 2, (size_t)B * H * head * 2, "split attention v^T plane");
    }
    wr(ctx3, (size_t)B * np * (mx_out ? 4 : 6) * H * 64, "split attention ctx planes");
    return hipSuccess;
}
hipError_t launch_text_attn(int dtype, const void* qkv, const float* rel_bias, const int64_t* mask, void* ctx, int T, int L, int H, hipStream_t, const unsigned* run_if, void* planes, unsigned* flag) {
    if (run_if) { rd(run_if, 4, "text attention predicate"); if (*run_if == 0) return hipSuccess; }
    if (flag) rd(flag, 4, "text guard word");
    if (planes) { if (dtype != DT_F32) return hipErrorInvalidValue; wr(planes, (size_t)T * L * 3 * H * 64 * 2, "text attention planes"); ctx = nullptr; }
    rd(qkv, (size_t)T * L * 3 * H * 64 * esz(dtype), "text attention qkv");
    rd(rel_bias, (size_t)H * L * L * 4, "text attention bias");
    rd(mask, (size_t)T * L * 8, "text attention mask");
    if (ctx) wr(ctx, (size_t)T * L * H * 64 * esz(dtype), "text attention ctx");
    return hipSuccess;
}

// ---- row kernels (rowops.hip) ------------------------------------------------------------------------------------------------
hipError_t launch_ln_finalize(const float* part, float* mu, float* stat, float, int64_t rows, hipStream_t, bool) {
    rd(part, (size_t)rows * 24 * 4, "ln_finalize partials"); wr(mu, (size_t)rows * 4, "ln_finalize mu"); wr(stat, (size_t)rows * 8, "ln_finalize stat");
    return hipSuccess;
}
hipError_t launch_ln_prepare(int dtype, const float* in, const float* g, const float* b, float, float* out_f32, const float* gain, void* copy_t, float* mu, float* stat, float, int64_t rows, int D, hipStream_t) {
    rd(in, (size_t)rows * D * 4, "ln_prepare in");
    if (g) { rd(g, (size_t)D * 4, "ln_prepare gamma"); rd(b, (size_t)D * 4, "ln_prepare beta"); wr(out_f32, (size_t)rows * D * 4, "ln_prepare out"); }
    rd(gain, (size_t)D * 4, "ln_prepare gain");
    wr(copy_t, (size_t)rows * D * esz(dtype), "ln_prepare copy"); wr(mu, (size_t)rows * 4, "ln_prepare mu"); wr(stat, (size_t)rows * 8, "ln_prepare stat");
    return hipSuccess;
}
hipError_t launch_layernorm_split3(const float* in, const float* g, const float* b, float, void* out3, int64_t rows, int D, unsigned*, hipStream_t, int mx, float* out_f32) {
    rd(in, (size_t)rows * D * 4, "layernorm_split3 in"); rd(g, (size_t)D * 4, "gamma"); rd(b, (size_t)D * 4, "beta");
    if (out_f32) wr(out_f32, (size_t)rows * D * 4, "layernorm_split3 fp32 out");
    wr(out3, (size_t)rows * (mx ? 4 : 6) * D, "layernorm_split3 planes");
    return hipSuccess;
}
hipError_t launch_layernorm(int dtype, const float* in, const float* g, const float* b, float, void* out_t, float* out_f32, int64_t rows, int D, hipStream_t, const unsigned* run_if) {
    if (run_if) { rd(run_if, 4, "layernorm predicate"); if (*run_if == 0) return hipSuccess; }
    rd(in, (size_t)rows * D * 4, "layernorm in"); rd(g, (size_t)D * 4, "gamma"); rd(b, (size_t)D * 4, "beta");
    if (out_t) wr(out_t, (size_t)rows * D * esz(dtype), "layernorm out_t");
    if (out_f32) wr(out_f32, (size_t)rows * D * 4, "layernorm out_f32");
    return hipSuccess;
}
hipError_t launch_im2col(int dtype, const float* px, void* out, int B, int C, int H, int W, int, int, int, int n_pad, int k_pad, hipStream_t, const unsigned* run_if) {
    if (run_if) { rd(run_if, 4, "im2col predicate"); if (*run_if == 0) return hipSuccess; }
    rd(px, (size_t)B * C * H * W * 4, "im2col pixels"); wr(out, (size_t)B * n_pad * k_pad * esz(dtype), "im2col matrix");
    return hipSuccess;
}
hipError_t launch_text_embed(int dtype, const int64_t* ids, const float* we, const float* pe, const float* g, const float* b, float, float* h, void* xn, int T, int L, int D, int vocab, int max_pos, int, hipStream_t, const unsigned* run_if) {
    if (run_if) { rd(run_if, 4, "text embed predicate"); if (*run_if == 0) return hipSuccess; }
    rd(ids, (size_t)T * L * 8, "text ids"); rd(we, (size_t)vocab * D * 4, "word embeddings"); rd(pe, (size_t)max_pos * D * 4, "position embeddings");
    rd(g, (size_t)D * 4, "gamma"); rd(b, (size_t)D * 4, "beta");
    wr(h, (size_t)T * L * D * 4, "text h"); wr(xn, (size_t)T * L * D * esz(dtype), "text xn");
    return hipSuccess;
}
hipError_t launch_masked_meanpool(const float* h, const int64_t* mask, float* out, int T, int L, int D, hipStream_t) {
    rd(h, (size_t)T * L * D * 4, "meanpool h"); rd(mask, (size_t)T * L * 8, "meanpool mask"); wr(out, (size_t)T * D * 4, "meanpool out");
    return hipSuccess;
}
hipError_t launch_rows_dot(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, float* out, int M, int N, int K, int rpg, int64_t og, int64_t orow,
                           int64_t ocol, hipStream_t) {
    rd(a, ((size_t)(M - 1) * lda + K) * 4, "rows_dot a"); rd(b, ((size_t)(N - 1) * ldb + K) * 4, "rows_dot b");
    if (bias) rd(bias, (size_t)N * 4, "rows_dot bias");
    const size_t last = (size_t)((M - 1) / rpg) * og + (size_t)std::min(M - 1, rpg - 1) * orow + (size_t)(N - 1) * ocol;
    wr(out, (last + 1) * 4, "rows_dot out");
    return hipSuccess;
}
hipError_t launch_image_features(const float* tokens, int64_t image_stride, int B, int n_tokens, int D, float* out, hipStream_t) {
    rd(tokens, ((size_t)(B - 1) * image_stride + n_tokens) * D * 4, "image_features tokens"); wr(out, (size_t)B * 2 * D * 4, "image_features out");
    return hipSuccess;
}
hipError_t launch_ln_l2norm(const float* in, int64_t ld, const float* g, const float* b, float, float* out, int64_t rows, int D, int, hipStream_t) {
    rd(in, ((size_t)(rows - 1) * ld + D) * 4, "ln_l2norm in"); rd(g, (size_t)D * 4, "gamma"); rd(b, (size_t)D * 4, "beta"); wr(out, (size_t)rows * D * 4, "ln_l2norm out");
    return hipSuccess;
}
hipError_t launch_copy_tokens(const float* src, float* dst, int B, int nv, int np, int D, hipStream_t, const unsigned* run_if) {
    if (run_if) { rd(run_if, 4, "copy_tokens predicate"); if (*run_if == 0) return hipSuccess; }
    rd(src, (size_t)B * np * D * 4, "copy_tokens src"); wr(dst, (size_t)B * nv * D * 4, "copy_tokens dst");
    return hipSuccess;
}

hipError_t launch_guard_word(unsigned* words, int op, hipStream_t, int flag_idx, int count_idx) {      // rowops.hip: the fp32 mode's overflow-guard words (8 x u32)
    if (!words || (op != 0 && op != 1) || flag_idx < 0 || flag_idx > 7 || count_idx < 0 || count_idx > 7) return hipErrorInvalidValue;
    rd(words, 32, "guard words"); wr(words, 32, "guard words");
    if (op == 0) words[flag_idx] = 0;
    else if (words[flag_idx]) words[count_idx] += 1;
    return hipSuccess;
}

// ---- VL-CABS + post-processing (vlcabs.hip) -----------------------------------------------------------------------------------
size_t vlcabs_workspace_floats(int B, int T, int n_pad, int D) { return (size_t)B * (n_pad / 128) * T * (D + 2); }      // vlcabs.hip:203
hipError_t launch_vlcabs(const float* tokens, const float* g, const float* b, float, const float* qhat, float, float, int, float* ws, float* scores, float* t2i, float* logits, int B, int T, int nv, int np, int D, hipStream_t) {
    rd(tokens, (size_t)B * np * D * 4, "vlcabs tokens"); rd(g, (size_t)D * 4, "gamma"); rd(b, (size_t)D * 4, "beta"); rd(qhat, (size_t)T * D * 4, "vlcabs queries");
    wr(ws, vlcabs_workspace_floats(B, T, np, D) * 4, "vlcabs workspace");
    wr(scores, (size_t)B * T * nv * 4, "vlcabs scores"); wr(t2i, (size_t)T * B * 4, "vlcabs t2i"); wr(logits, (size_t)B * T * 4, "vlcabs logits");
    return hipSuccess;
}
hipError_t launch_upsample_bilinear(const float* maps, int64_t stride, float* out, int64_t* amax, int M, int g, int Ho, int Wo, int, int, hipStream_t) {
    rd(maps, ((size_t)(M - 1) * stride + (size_t)g * g) * 4, "upsample maps"); wr(out, (size_t)M * Ho * Wo * 4, "upsample out");
    if (amax) wr(amax, (size_t)M * 8, "upsample argmax");
    return hipSuccess;
}
hipError_t launch_grounding_points(const float* maps, int64_t stride, unsigned long long* keys, int* xy, int M, int g, int, int, int, hipStream_t) {
    rd(maps, ((size_t)(M - 1) * stride + (size_t)g * g) * 4, "grounding maps"); wr(keys, (size_t)M * 8, "grounding keys"); wr(xy, (size_t)M * 2 * 4, "grounding xy");
    return hipSuccess;
}

// ---- preprocessing (preprocess.hip) -------------------------------------------------------------------------------------------
struct PreDescMock { const void* img; int dtype, H, W, C; int pad_left, pad_top, PH, PW; const int* bounds_h; const int* kk_h; int ksize_h; const int* bounds_v; const int* kk_v; int ksize_v; int64_t a8, b8, c8; };
size_t preprocess_batch_desc_bytes(int n) { return ((size_t)n * sizeof(PreDescMock) + 8 * (size_t)n + 255) / 256 * 256; }
hipError_t launch_preprocess_batch(const void* descs_host, int n, int max_ph, int S, const float* mean, const float* stdv, float, unsigned char* ws, float* out, int, hipStream_t) {
    rd(descs_host, (size_t)n * sizeof(PreDescMock), "preprocess descriptors (host)");
    rd(mean, 12, "mean"); rd(stdv, 12, "std");
    wr(ws, preprocess_batch_desc_bytes(n), "preprocess descriptor block");
    const PreDescMock* d = (const PreDescMock*)descs_host;
    for (int i = 0; i < n; ++i) {
        const size_t px = d[i].dtype == 0 ? 1 : d[i].dtype == 1 ? 2 : 4;
        rd(d[i].img, (size_t)d[i].H * d[i].W * d[i].C * px, "raw image");
        rd(d[i].bounds_h, (size_t)S * 2 * 4, "bounds_h"); rd(d[i].kk_h, (size_t)S * d[i].ksize_h * 4, "kk_h");
        rd(d[i].bounds_v, (size_t)S * 2 * 4, "bounds_v"); rd(d[i].kk_v, (size_t)S * d[i].ksize_v * 4, "kk_v");
        wr(ws + d[i].a8, (size_t)d[i].PH * d[i].PW * d[i].C, "8-bit padded image");
        wr(ws + d[i].b8, (size_t)d[i].PH * S * d[i].C, "horizontally resampled image");
        wr(ws + d[i].c8, (size_t)S * S * d[i].C, "resampled image");
        if (d[i].PH > max_ph) { fprintf(stderr, "[host_asan] max_ph too small\n"); abort(); }
    }
    wr(out, (size_t)n * 3 * S * S * 4, "pixel_values");
    return hipSuccess;
}
hipError_t launch_preprocess(const void* img, int dt, int H, int W, int C, int S, const int* bh, const int* kh, int ksh, const int* bv, const int* kv, int ksv, const float* mean, const float* stdv, float, unsigned char* ws8, unsigned* mm, float* out, int, hipStream_t) {
    rd(img, (size_t)H * W * C * (dt == 0 ? 1 : dt == 1 ? 2 : 4), "raw image");
    rd(bh, (size_t)S * 8, "bounds_h"); rd(kh, (size_t)S * ksh * 4, "kk_h"); rd(bv, (size_t)S * 8, "bounds_v"); rd(kv, (size_t)S * ksv * 4, "kk_v");
    rd(mean, 12, "mean"); rd(stdv, 12, "std");
    wr(ws8, (size_t)H * W * C + (size_t)H * S * C + (size_t)S * S * C, "preprocess scratch"); wr(mm, 8, "min/max words"); wr(out, (size_t)3 * S * S * 4, "pixel_values");
    return hipSuccess;
}

}  // namespace rz
