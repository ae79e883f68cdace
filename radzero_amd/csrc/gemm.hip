// radzero_hip — tiled MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogues), gfx950.
//
// Every linear layer on the path is y = x W^T + b with x row-major [M,K] and W row-major [N,K]
// (nn.Linear storage), so BOTH operands are K-contiguous: exactly the MFMA fragment shape.
// Replaces (TF: = transformers/models): TF:dinov2/modeling_dinov2.py:199-213 (q/k/v), :246-251
// (attention.output.dense), :281-297 (fc1/GELU/fc2), :272-278 (LayerScale) and the residual adds of
// :342-380; TF:mpnet/modeling_mpnet.py:131-171,:204-231; the patch-embedding conv
// TF:dinov2/modeling_dinov2.py:139-148 as an im2col GEMM.
//
// Structure (v1): 128x128 output tile, 4 waves (2x2, 64x64 each = 4x4 MFMA 16x16 tiles), K panel of
// 128 bytes per step (64 x 16-bit or 32 x f32), two LDS stages filled by global_load_lds_dwordx4 with
// the panel XOR swizzle of rz_common.h.  M must be a multiple of 128 (callers pad rows per image),
// N a multiple of 128, K*sizeof(T) a multiple of 128.
#include "rz_common.h"
#include "rz_kernels.h"

namespace rz {

constexpr int BM = 128, BN = 128;
constexpr int PANEL_BYTES = 128 * 128;  // one operand tile in LDS (128 rows x 128 B)

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[4 * PANEL_BYTES];  // A0 A1 B0 B1
    constexpr bool SWAP = (EPI != EPI_VT);
    constexpr int KS = 128 / (32 * (int)sizeof(T));  // MFMA k-steps (of 32 elements) per panel
    typedef typename Traits<T>::frag frag_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    // tile mapping: n-tiles fastest so that the blocks sharing an A row-panel run together;
    // XCD remap keeps such a run on one XCD's L2.
    const int tiles_n = g.N / BN;
    const int ntiles = (g.M / BM) * tiles_n;
    const int bid = xcd_remap(blockIdx.x, ntiles);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

    const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * sizeof(T);
    const char* Wb = reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * sizeof(T);
    const int64_t lda_b = g.lda * (int64_t)sizeof(T), ldw_b = g.ldw * (int64_t)sizeof(T);
    const int nk = (g.K * (int)sizeof(T)) / 128;

    auto stage = [&](int kt, int buf) {
        char* sa = lds + buf * PANEL_BYTES;
        char* sb = lds + (2 + buf) * PANEL_BYTES;
        const char* ga = Ab + (int64_t)kt * 128;
        const char* gb = Wb + (int64_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row8 = (wave * 4 + i) * 8;
            glds_rows8(sa + row8 * 128, ga, lda_b, row8, lane);
            glds_rows8(sb + row8 * 128, gb, ldw_b, row8, lane);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* sa = lds + buf * PANEL_BYTES;
        const char* sb = lds + (2 + buf) * PANEL_BYTES;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            frag_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = lds_frag<T>(sa, wm * 64 + i * 16 + l15, ks * 4 + lg);
                fb[i] = lds_frag<T>(sb, wn * 64 + i * 16 + l15, ks * 4 + lg);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (SWAP) acc[i][j] = mma(fb[j], fa[i], acc[i][j]);   // D[row=n][col=m]
                    else acc[i][j] = mma(fa[i], fb[j], acc[i][j]);        // D[row=m][col=n]
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---------------- epilogue ----------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = acc[i][j];
            if constexpr (SWAP) {
                const int m = m0 + wm * 64 + i * 16 + l15;
                const int n = n0 + wn * 64 + j * 16 + 4 * lg;     // 4 consecutive columns n..n+3
                f32x4 v = a;
                if (g.bias) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n);
                    v += b;
                }
                if constexpr (EPI == EPI_STORE) {
                    T* o = reinterpret_cast<T*>(g.out) + (int64_t)m * g.ldo + n;
                    *reinterpret_cast<typename Traits<T>::vec4*>(o) = pack4<T>(v[0], v[1], v[2], v[3]);
                } else if constexpr (EPI == EPI_GELU) {
                    T* o = reinterpret_cast<T*>(g.out) + (int64_t)m * g.ldo + n;
                    *reinterpret_cast<typename Traits<T>::vec4*>(o) =
                        pack4<T>(gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3]));
                } else if constexpr (EPI == EPI_HEADS) {
                    // out[b][head][tok][64], head = n/64 over `heads_total` heads (q heads then k heads)
                    const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                    T* o = reinterpret_cast<T*>(g.out) +
                           (((int64_t)b * g.heads_total + (n >> 6)) * g.rows_per_image + tok) * 64 + (n & 63);
                    *reinterpret_cast<typename Traits<T>::vec4*>(o) = pack4<T>(v[0], v[1], v[2], v[3]);
                } else if constexpr (EPI == EPI_RESID_SCALE) {
                    // h[m][n] += lambda[n] * (acc + bias[n])   (fp32 residual stream, in place)
                    const f32x4 s = *reinterpret_cast<const f32x4*>(g.scale + n);
                    float* r = g.resid + (int64_t)m * g.ldr + n;
                    f32x4 h = *reinterpret_cast<f32x4*>(r);
                    h += s * v;
                    *reinterpret_cast<f32x4*>(r) = h;
                } else if constexpr (EPI == EPI_RESID_ADD) {
                    // out_f32[m][n] = acc + bias[n] + resid[m][n]   (post-LN blocks: LN applied by the next kernel)
                    const f32x4 h = *reinterpret_cast<const f32x4*>(g.resid + (int64_t)m * g.ldr + n);
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + n) = v + h;
                } else if constexpr (EPI == EPI_PATCH) {
                    // h[m][n] = acc + posb[tok][n]; posb = pos-embed + (cls | conv bias), zero on pad rows
                    const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                    const f32x4 p = *reinterpret_cast<const f32x4*>(g.scale + (int64_t)tok * g.N + n);
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + n) = v + p;
                } else if constexpr (EPI == EPI_STORE_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + n) = v;
                }
            } else {
                // EPI_VT: lane owns column n, rows m..m+3 (4 consecutive tokens of one image)
                const int m = m0 + wm * 64 + i * 16 + 4 * lg;
                const int n = n0 + wn * 64 + j * 16 + l15;
                const float bv = g.bias ? g.bias[n] : 0.f;
                const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                // vT[b][head][d][tok]
                T* o = reinterpret_cast<T*>(g.out) +
                       (((int64_t)b * g.heads_total + (n >> 6)) * 64 + (n & 63)) * g.rows_per_image + tok;
                *reinterpret_cast<typename Traits<T>::vec4*>(o) = pack4<T>(a[0] + bv, a[1] + bv, a[2] + bv, a[3] + bv);
            }
        }
    }
}

template <typename T>
static hipError_t launch_gemm_t(int epi, const GemmArgs& g, hipStream_t s) {
    const int ntiles = (g.M / BM) * (g.N / BN);
    dim3 grid(ntiles), block(256);
#define RZ_CASE(E) \
    case E: hipLaunchKernelGGL((gemm_kernel<T, E>), grid, block, 0, s, g); break;
    switch (epi) {
        RZ_CASE(EPI_STORE)
        RZ_CASE(EPI_GELU)
        RZ_CASE(EPI_HEADS)
        RZ_CASE(EPI_VT)
        RZ_CASE(EPI_RESID_SCALE)
        RZ_CASE(EPI_RESID_ADD)
        RZ_CASE(EPI_PATCH)
        RZ_CASE(EPI_STORE_F32)
        default: return hipErrorInvalidValue;
    }
#undef RZ_CASE
    return hipGetLastError();
}

hipError_t launch_gemm(int dtype, int epi, const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return hipErrorInvalidValue;
    if (g.M % BM || g.N % BN) return hipErrorInvalidValue;
    const int esz = dtype == DT_F32 ? 4 : 2;
    if ((g.K * esz) % 128) return hipErrorInvalidValue;
    if ((g.lda * esz) % 16 || (g.ldw * esz) % 16) return hipErrorInvalidValue;
    switch (dtype) {
        case DT_F32: return launch_gemm_t<float>(epi, g, s);
        case DT_BF16: return launch_gemm_t<bf16_t>(epi, g, s);
        case DT_F16: return launch_gemm_t<f16_t>(epi, g, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace rz
