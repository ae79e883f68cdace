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

// Tile rasterisation.  Each XCD (private 4 MB L2) runs a contiguous range of logical tile ids (xcd_remap);
// within that range ids walk GROUP_M m-tiles, then step to the next n-tile, so the ~32-64 tiles in flight
// on an XCD form a compact (GROUP_M x 8) block that shares its A and W panels through L2 instead of
// re-fetching them from the Infinity Cache / HBM (128x128 tiles alone are L2-bandwidth bound otherwise).
template <int GROUP_M>
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int per_group = GROUP_M * tiles_n;
    const int grp = id / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int r = id - grp * per_group;
    tm = first_m + r % gsz;
    tn = r / gsz;
}

// Fused epilogue for one wave's 64x64 accumulator block (4x4 MFMA tiles); (mw, nw) = block origin.
template <typename T, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x4 (&acc)[4][4], int mw, int nw, int l15, int lg) {
    constexpr bool SWAP = (EPI != EPI_VT);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = acc[i][j];
            if constexpr (SWAP) {
                const int m = mw + i * 16 + l15;
                const int n = nw + j * 16 + 4 * lg;     // 4 consecutive columns n..n+3
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
                        pack4<T>(gelu_for<T>(v[0]), gelu_for<T>(v[1]), gelu_for<T>(v[2]), gelu_for<T>(v[3]));
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
                const int m = mw + i * 16 + 4 * lg;
                const int n = nw + j * 16 + l15;
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

// ---------------------------------------------------------------------------------------------------
// LDS-staged epilogue (256x256 kernels).  Measured with the debug flags of tools/kbench.py: the direct epilogue above
// (8/16-byte pieces, 32-64 B per row per instruction) costs 31 % of a K=768 GEMM and nothing overlaps it (one
// workgroup per CU).  Here each wave drops a 64x64 fp32 block of accumulators into its private 16 KB LDS region
// (XOR-swizzled 16-B chunks), reads it back row-major and touches global memory in full lines:
// 4 rows x 256 B (fp32 read-modify-write) or 4 rows x 128 B (16-bit stores) per wave instruction.
// `outer`/`inner`: SWAP epilogues outer = m, inner = n; EPI_VT outer = n, inner = m (so V^T rows are written whole).
// ---------------------------------------------------------------------------------------------------
template <typename T, int EPI>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs& g, const f32x4 (&acc)[4][4], char* wlds, int mw, int nw, int lane) {
    constexpr bool SWAP = (EPI != EPI_VT);
    const int l15 = lane & 15, lg = lane >> 4;
    // write: lane holds, for tile (i, j), 4 consecutive inner indices of one outer index
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int outer = (SWAP ? i : j) * 16 + l15;
            const int chunk = (SWAP ? j : i) * 4 + lg;                 // 16-B chunk index along inner (0..15)
            *reinterpret_cast<f32x4*>(wlds + outer * 256 + ((chunk ^ (outer & 15)) << 4)) = acc[i][j];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // same wave, in-order LDS queue; keeps hipcc from reordering
    // 16-bit row-major outputs: 8 values = ONE 16-byte store per lane, 8 rows x 128 B per wave instruction (the epilogue is
    // store-ISSUE bound: half the store instructions of the 8-byte form)
    if constexpr (sizeof(T) == 2 && (EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_HEADS || EPI == EPI_VT)) {
        const int c2 = (lane & 7) * 2;                                     // first of two 16-B fp32 chunks = 8 inner indices
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            const int outer = it * 8 + (lane >> 3);
            f32x4 v0 = *reinterpret_cast<const f32x4*>(wlds + outer * 256 + ((c2 ^ (outer & 15)) << 4));
            f32x4 v1 = *reinterpret_cast<const f32x4*>(wlds + outer * 256 + (((c2 + 1) ^ (outer & 15)) << 4));
            T* o;
            if constexpr (EPI == EPI_VT) {
                // outer = feature n, inner = 8 consecutive tokens of one image: vT[b][head][d][tok..tok+7]
                const int n = nw + outer, m = mw + c2 * 4;
                const float bv = g.bias ? g.bias[n] : 0.f;
                v0 += bv; v1 += bv;
                const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                o = reinterpret_cast<T*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * 64 + (n & 63)) * g.rows_per_image + tok;
            } else {
                const int m = mw + outer, n = nw + c2 * 4;
                if (g.bias) { v0 += *reinterpret_cast<const f32x4*>(g.bias + n); v1 += *reinterpret_cast<const f32x4*>(g.bias + n + 4); }
                if constexpr (EPI == EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = gelu_for<T>(v0[e]); v1[e] = gelu_for<T>(v1[e]); }
                }
                if constexpr (EPI == EPI_HEADS) {
                    const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                    o = reinterpret_cast<T*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * g.rows_per_image + tok) * 64 + (n & 63);
                } else {
                    o = reinterpret_cast<T*>(g.out) + (int64_t)m * g.ldo + n;
                }
            }
            *reinterpret_cast<typename Traits<T>::frag*>(o) = pack8<T>(v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }
    const int c = lane & 15;                                           // chunk read by this lane
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int outer = it * 4 + (lane >> 4);
        f32x4 v = *reinterpret_cast<const f32x4*>(wlds + outer * 256 + ((c ^ (outer & 15)) << 4));
        const int inner = c * 4;
        const int m = mw + (SWAP ? outer : inner);
        const int n = nw + (SWAP ? inner : outer);
        if constexpr (SWAP) {
            if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + n);
            if constexpr (EPI == EPI_STORE) {
                *reinterpret_cast<typename Traits<T>::vec4*>(reinterpret_cast<T*>(g.out) + (int64_t)m * g.ldo + n) = pack4<T>(v[0], v[1], v[2], v[3]);
            } else if constexpr (EPI == EPI_GELU) {
                *reinterpret_cast<typename Traits<T>::vec4*>(reinterpret_cast<T*>(g.out) + (int64_t)m * g.ldo + n) =
                    pack4<T>(gelu_for<T>(v[0]), gelu_for<T>(v[1]), gelu_for<T>(v[2]), gelu_for<T>(v[3]));
            } else if constexpr (EPI == EPI_HEADS) {
                const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                T* o = reinterpret_cast<T*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * g.rows_per_image + tok) * 64 + (n & 63);
                *reinterpret_cast<typename Traits<T>::vec4*>(o) = pack4<T>(v[0], v[1], v[2], v[3]);
            } else if constexpr (EPI == EPI_RESID_SCALE) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(g.scale + n);
                float* r = g.resid + (int64_t)m * g.ldr + n;
                *reinterpret_cast<f32x4*>(r) = *reinterpret_cast<const f32x4*>(r) + sc * v;
            } else if constexpr (EPI == EPI_RESID_ADD) {
                const f32x4 h = *reinterpret_cast<const f32x4*>(g.resid + (int64_t)m * g.ldr + n);
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + n) = v + h;
            } else if constexpr (EPI == EPI_PATCH) {
                const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                const f32x4 p = *reinterpret_cast<const f32x4*>(g.scale + (int64_t)tok * g.N + n);
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + n) = v + p;
            } else if constexpr (EPI == EPI_STORE_F32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + n) = v;
            }
        } else {
            // EPI_VT: one feature n, tokens m..m+3 of one image: vT[b][head][d][tok]
            const float bv = g.bias ? g.bias[n] : 0.f;
            const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
            T* o = reinterpret_cast<T*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * 64 + (n & 63)) * g.rows_per_image + tok;
            *reinterpret_cast<typename Traits<T>::vec4*>(o) = pack4<T>(v[0] + bv, v[1] + bv, v[2] + bv, v[3] + bv);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[4 * PANEL_BYTES];  // A0 A1 B0 B1
    constexpr bool SWAP = (EPI != EPI_VT);
    constexpr int KS = 128 / (32 * (int)sizeof(T));  // MFMA k-steps (of 32 elements) per panel
    typedef typename Traits<T>::frag frag_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    const int tiles_n = g.N / BN, tiles_m = g.M / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    tile_coords<8>(bid, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

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
        // fragments of k-step ks+1 are requested before the 16 MFMAs of k-step ks (register double buffer),
        // so only the first LDS round trip of a K panel is exposed
        frag_t fa[2][4], fb[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[0][i] = lds_frag<T>(sa, wm * 64 + i * 16 + l15, lg);
            fb[0][i] = lds_frag<T>(sb, wn * 64 + i * 16 + l15, lg);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[(ks + 1) & 1][i] = lds_frag<T>(sa, wm * 64 + i * 16 + l15, (ks + 1) * 4 + lg);
                    fb[(ks + 1) & 1][i] = lds_frag<T>(sb, wn * 64 + i * 16 + l15, (ks + 1) * 4 + lg);
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (SWAP) acc[i][j] = mma(fb[ks & 1][j], fa[ks & 1][i], acc[i][j]);   // D[row=n][col=m]
                    else acc[i][j] = mma(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);        // D[row=m][col=n]
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    gemm_epilogue<T, EPI>(g, acc, m0 + wm * 64, n0 + wn * 64, l15, lg);
}

// ---------------------------------------------------------------------------------------------------
// v2: 256x128 tile, 8 waves (4x2, 64x64 each), THREE LDS stages of 48 KB and counted vmcnt: tile kt+2 is
// requested before tile kt is computed, and the end-of-step wait retires only tile kt+1
// (s_waitcnt vmcnt(6) = the 6 global_load_lds of tile kt+2 may stay in flight across the raw s_barrier).
// One block per CU (144 KB LDS), 2 waves per SIMD.  Requires M % 256 == 0.
// ---------------------------------------------------------------------------------------------------
constexpr int BM2 = 256;
constexpr int STAGE2_BYTES = (BM2 + BN) * 128;   // A panel 32 KB + B panel 16 KB

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v2(GemmArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[3 * STAGE2_BYTES];
    constexpr bool SWAP = (EPI != EPI_VT);
    constexpr int KS = 128 / (32 * (int)sizeof(T));
    typedef typename Traits<T>::frag frag_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    const int tiles_n = g.N / BN, tiles_m = g.M / BM2;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    tile_coords<4>(bid, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM2, n0 = tn * BN;

    const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * sizeof(T);
    const char* Wb = reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * sizeof(T);
    const int64_t lda_b = g.lda * (int64_t)sizeof(T), ldw_b = g.ldw * (int64_t)sizeof(T);
    const int nk = (g.K * (int)sizeof(T)) / 128;

    auto stage = [&](int kt, int buf) {      // 6 global_load_lds_dwordx4 per wave
        char* sa = lds + buf * STAGE2_BYTES;
        char* sb = sa + BM2 * 128;
        const char* ga = Ab + (int64_t)kt * 128;
        const char* gb = Wb + (int64_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row8 = (wave * 4 + i) * 8;
            glds_rows8(sa + row8 * 128, ga, lda_b, row8, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row8 = (wave * 2 + i) * 8;
            glds_rows8(sb + row8 * 128, gb, ldw_b, row8, lane);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    if (nk > 1) {
        stage(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 2 < nk;
        if (more) stage(kt + 2, buf >= 1 ? buf - 1 : 2);      // (buf + 2) % 3
        const char* sa = lds + buf * STAGE2_BYTES;
        const char* sb = sa + BM2 * 128;
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
                    if (SWAP) acc[i][j] = mma(fb[j], fa[i], acc[i][j]);
                    else acc[i][j] = mma(fa[i], fb[j], acc[i][j]);
                }
        }
        // retire tile kt+1 (this wave's share); tile kt+2 stays in flight across the barrier
        if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf = buf == 2 ? 0 : buf + 1;
    }
    gemm_epilogue<T, EPI>(g, acc, m0 + wm * 64, n0 + wn * 64, l15, lg);
}

// ---------------------------------------------------------------------------------------------------
// v3: 256x256 tile, 8 waves (2x4), each wave 128x64 (8x4 MFMA tiles, 128 accumulator VGPRs), two LDS stages
// of 64 KB.  Halves the L2->LDS operand traffic per FLOP of the 128x128 kernel (which is what bounds it:
// ~12 TB/s of L2 reads at 86 % hit rate) and lowers LDS reads per MFMA from 0.5 to 0.375.
// Requires M % 256 == 0 and N % 256 == 0.
// ---------------------------------------------------------------------------------------------------
constexpr int BN3 = 256;
constexpr int STAGE3_BYTES = (BM2 + BN3) * 128;   // 64 KB

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v3(GemmArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE3_BYTES];
    constexpr bool SWAP = (EPI != EPI_VT);
    constexpr int KS = 128 / (32 * (int)sizeof(T));
    typedef typename Traits<T>::frag frag_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    const int tiles_n = g.N / BN3, tiles_m = g.M / BM2;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    tile_coords<4>(bid, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM2, n0 = tn * BN3;

    const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * sizeof(T);
    const char* Wb = reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * sizeof(T);
    const int64_t lda_b = g.lda * (int64_t)sizeof(T), ldw_b = g.ldw * (int64_t)sizeof(T);
    const int nk = (g.K * (int)sizeof(T)) / 128;

    auto stage = [&](int kt, int buf) {      // 8 global_load_lds_dwordx4 per wave
        char* sa = lds + buf * STAGE3_BYTES;
        char* sb = sa + BM2 * 128;
        const char* ga = Ab + (int64_t)kt * 128;
        const char* gb = Wb + (int64_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row8 = (wave * 4 + i) * 8;
            glds_rows8(sa + row8 * 128, ga, lda_b, row8, lane);
            if (!(g.debug_flags & 2)) glds_rows8(sb + row8 * 128, gb, ldw_b, row8, lane);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* sa = lds + buf * STAGE3_BYTES;
        const char* sb = sa + BM2 * 128;
        if (!(g.debug_flags & 1))
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            frag_t fa[8], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i] = lds_frag<T>(sb, wn * 64 + i * 16 + l15, ks * 4 + lg);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = lds_frag<T>(sa, wm * 128 + i * 16 + l15, ks * 4 + lg);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (SWAP) acc[i][j] = mma(fb[j], fa[i], acc[i][j]);
                    else acc[i][j] = mma(fa[i], fb[j], acc[i][j]);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (g.debug_flags & 4) return;      // measurement only: no epilogue
    const f32x4 (&lo)[4][4] = *reinterpret_cast<const f32x4 (*)[4][4]>(&acc[0]);
    const f32x4 (&hi)[4][4] = *reinterpret_cast<const f32x4 (*)[4][4]>(&acc[4]);
    // default: LDS-staged 16-byte stores for the 16-bit row-major / per-head / transposed outputs (+9..15 % on those GEMMs),
    // direct epilogue for GELU (VALU-bound) and the fp32 residual read-modify-write (equal within noise).
    // debug bit3 forces the LDS-staged epilogue everywhere, bit4 the direct one everywhere.
    constexpr bool kLdsDefault = sizeof(T) == 2 && (EPI == EPI_STORE || EPI == EPI_HEADS || EPI == EPI_VT);
    if ((g.debug_flags & 16) || (!kLdsDefault && !(g.debug_flags & 8))) {
        gemm_epilogue<T, EPI>(g, lo, m0 + wm * 128, n0 + wn * 64, l15, lg);
        gemm_epilogue<T, EPI>(g, hi, m0 + wm * 128 + 64, n0 + wn * 64, l15, lg);
        return;
    }
    __syncthreads();                    // every wave is done reading operand stages: LDS is free
    char* wlds = lds + wave * (64 * 256);
    gemm_epilogue_lds<T, EPI>(g, lo, wlds, m0 + wm * 128, n0 + wn * 64, lane);
    gemm_epilogue_lds<T, EPI>(g, hi, wlds, m0 + wm * 128 + 64, n0 + wn * 64, lane);
}

// ---------------------------------------------------------------------------------------------------
// v4 (16-bit operands): 256x256 tile, 8 waves (2x4, 128x64 each), K step of 32 elements (64-byte rows) and a
// FOUR-stage LDS ring (4 x 32 KB) with counted vmcnt: three stages are in flight while one is computed and the
// stream never drains inside a tile.  Why: measured on MI355X (tools/mb_ldsdma.hip) one CU moves at most
// ~45-50 GB/s L2->LDS by global_load_lds whatever the ring depth, so operand staging (64 KB per 64-deep K step of
// a 256x256 tile = 1.29 us) is the real bound of these GEMMs; it has to overlap the MFMA work completely.
// LDS image: two 64-B tile rows per 128-B line; 16-B chunk c of row r sits at chunk position
// (((r&1)<<2)|c) ^ f(r), f(r) = ((r&1)<<1) | (((r>>2)&1)<<2)  — conflict-free for the ds_read_b128 fragment
// pattern (16 rows x 1 chunk per lane group; found by exhaustive search over XOR-linear maps).
// ---------------------------------------------------------------------------------------------------
constexpr int STAGE4_BYTES = (BM2 + BN3) * 64;   // 32 KB
constexpr int RING4 = 4;

__device__ __forceinline__ int half_off(int r, int c) {   // byte offset of chunk c (0..3) of 64-B row r
    const int f = ((r & 1) << 1) | (((r >> 2) & 1) << 2);
    return (r >> 1) * 128 + (((((r & 1) << 2) | c) ^ f) << 4);
}
// one wave fills 16 consecutive 64-B rows (1 KiB); row16 multiple of 16
__device__ __forceinline__ void glds_rows16_half(char* lds_wave_base, const char* gsrc_row0, int64_t ld_bytes, int row16, int lane) {
    const int R = (row16 >> 1) + (lane >> 3);
    const int hb = (R >> 1) & 1;
    const int b = ((lane >> 2) & 1) ^ hb;
    const int r = 2 * R + b;
    const int f = (b << 1) | (hb << 2);
    const int c = ((lane & 7) ^ f) & 3;
    const char* src = gsrc_row0 + (int64_t)r * ld_bytes + (c << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v4(GemmArgs g) {
    static_assert(sizeof(T) == 2, "v4 is for 16-bit operands");
    __shared__ __attribute__((aligned(1024))) char lds[RING4 * STAGE4_BYTES];
    constexpr bool SWAP = (EPI != EPI_VT);
    typedef typename Traits<T>::frag frag_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    const int tiles_n = g.N / BN3, tiles_m = g.M / BM2;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    tile_coords<4>(bid, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM2, n0 = tn * BN3;

    const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * 2;
    const char* Wb = reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * 2;
    const int64_t lda_b = g.lda * 2, ldw_b = g.ldw * 2;
    const int nk = g.K / 32;

    auto stage = [&](int kt, int slot) {      // 4 global_load_lds_dwordx4 per wave: 2 for A, 2 for W
        char* sa = lds + slot * STAGE4_BYTES;
        char* sb = sa + BM2 * 64;
        const char* ga = Ab + (int64_t)kt * 64;
        const char* gb = Wb + (int64_t)kt * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row16 = (wave * 2 + i) * 16;
            glds_rows16_half(sa + row16 * 64, ga, lda_b, row16, lane);
            glds_rows16_half(sb + row16 * 64, gb, ldw_b, row16, lane);
        }
    };
    // per-lane fragment offsets (row = 16*i + l15 -> only l15 enters the swizzle)
    const int foff = half_off(l15, lg);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: three stages in flight
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 2) stage(2, 2);

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // retire stage kt: the (up to two) younger stages stay in flight across the barrier
        const int younger = nk - 1 - kt;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < nk) stage(kt + 3, (slot + 3) & 3);      // slot of stage kt-1: every wave is past it
        const char* sa = lds + slot * STAGE4_BYTES + wm * (128 * 64);
        const char* sb = lds + slot * STAGE4_BYTES + BM2 * 64 + wn * (64 * 64);
        if (!(g.debug_flags & 1)) {
            frag_t fa[8], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const frag_t*>(sb + i * (16 * 64) + foff);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const frag_t*>(sa + i * (16 * 64) + foff);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (SWAP) acc[i][j] = mma(fb[j], fa[i], acc[i][j]);
                    else acc[i][j] = mma(fa[i], fb[j], acc[i][j]);
                }
        }
        slot = (slot + 1) & 3;
    }
    const f32x4 (&lo)[4][4] = *reinterpret_cast<const f32x4 (*)[4][4]>(&acc[0]);
    const f32x4 (&hi)[4][4] = *reinterpret_cast<const f32x4 (*)[4][4]>(&acc[4]);
    // default: LDS-staged 16-byte stores for the 16-bit row-major / per-head / transposed outputs (+9..15 % on those GEMMs),
    // direct epilogue for GELU (VALU-bound) and the fp32 residual read-modify-write (equal within noise).
    // debug bit3 forces the LDS-staged epilogue everywhere, bit4 the direct one everywhere.
    constexpr bool kLdsDefault = sizeof(T) == 2 && (EPI == EPI_STORE || EPI == EPI_HEADS || EPI == EPI_VT);
    if ((g.debug_flags & 16) || (!kLdsDefault && !(g.debug_flags & 8))) {
        gemm_epilogue<T, EPI>(g, lo, m0 + wm * 128, n0 + wn * 64, l15, lg);
        gemm_epilogue<T, EPI>(g, hi, m0 + wm * 128 + 64, n0 + wn * 64, l15, lg);
        return;
    }
    __syncthreads();                    // every wave is done reading operand stages: LDS is free
    char* wlds = lds + wave * (64 * 256);
    gemm_epilogue_lds<T, EPI>(g, lo, wlds, m0 + wm * 128, n0 + wn * 64, lane);
    gemm_epilogue_lds<T, EPI>(g, hi, wlds, m0 + wm * 128 + 64, n0 + wn * 64, lane);
}

// ---------------------------------------------------------------------------------------------------
// v5 (16-bit operands): 256x128 tile, FOUR waves (2x2, 128x64 each), K step of 32 (64-byte rows), three-stage ring
// (3 x 24 KB) with counted vmcnt — sized so that TWO workgroups share a CU (72 KB LDS, <=256 VGPRs at 2 waves/SIMD).
// Why: the phase decomposition of v3 (debug flags, tools/kbench.py) shows staging (0.10 ms), MFMA work (0.08 ms) and
// epilogue (0.08 ms) of a K=768 GEMM adding up serially because ONE workgroup owns the CU; with two independent
// workgroups one's epilogue / load waits overlap the other's MFMA work.
// ---------------------------------------------------------------------------------------------------
constexpr int STAGE5_BYTES = (BM2 + BN) * 64;   // 24 KB
constexpr int RING5 = 3;

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel_v5(GemmArgs g) {
    static_assert(sizeof(T) == 2, "v5 is for 16-bit operands");
    __shared__ __attribute__((aligned(1024))) char lds[RING5 * STAGE5_BYTES];
    constexpr bool SWAP = (EPI != EPI_VT);
    typedef typename Traits<T>::frag frag_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    const int tiles_n = g.N / BN, tiles_m = g.M / BM2;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    tile_coords<4>(bid, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM2, n0 = tn * BN;

    const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * 2;
    const char* Wb = reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * 2;
    const int64_t lda_b = g.lda * 2, ldw_b = g.ldw * 2;
    const int nk = g.K / 32;

    auto stage = [&](int kt, int slot) {      // 6 global_load_lds_dwordx4 per wave: 4 for A (256 rows), 2 for W (128 rows)
        char* sa = lds + slot * STAGE5_BYTES;
        char* sb = sa + BM2 * 64;
        const char* ga = Ab + (int64_t)kt * 64;
        const char* gb = Wb + (int64_t)kt * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row16 = (wave * 4 + i) * 16;
            glds_rows16_half(sa + row16 * 64, ga, lda_b, row16, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row16 = (wave * 2 + i) * 16;
            glds_rows16_half(sb + row16 * 64, gb, ldw_b, row16, lane);
        }
    };
    const int foff = half_off(l15, lg);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    if (nk > 1) stage(1, 1);

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // retire stage kt; the younger stage (if any) stays in flight across the barrier
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) stage(kt + 2, slot == 0 ? 2 : slot - 1);      // slot of stage kt-1: every wave is past it
        const char* sa = lds + slot * STAGE5_BYTES + wm * (128 * 64);
        const char* sb = lds + slot * STAGE5_BYTES + BM2 * 64 + wn * (64 * 64);
        if (!(g.debug_flags & 1)) {
            frag_t fa[8], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const frag_t*>(sb + i * (16 * 64) + foff);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const frag_t*>(sa + i * (16 * 64) + foff);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (SWAP) acc[i][j] = mma(fb[j], fa[i], acc[i][j]);
                    else acc[i][j] = mma(fa[i], fb[j], acc[i][j]);
                }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (g.debug_flags & 4) return;
    const f32x4 (&lo)[4][4] = *reinterpret_cast<const f32x4 (*)[4][4]>(&acc[0]);
    const f32x4 (&hi)[4][4] = *reinterpret_cast<const f32x4 (*)[4][4]>(&acc[4]);
    gemm_epilogue<T, EPI>(g, lo, m0 + wm * 128, n0 + wn * 64, l15, lg);
    gemm_epilogue<T, EPI>(g, hi, m0 + wm * 128 + 64, n0 + wn * 64, l15, lg);
}

static int g_debug_flags = 0;
void gemm_set_debug_flags(int f) { g_debug_flags = f; }
static int g_variant = 0;      // 0 = auto, 1/2/3 = force that kernel where its shape constraints hold
void gemm_force_v1(bool on) { g_variant = on ? 1 : 0; }
void gemm_set_variant(int v) { g_variant = v; }

template <typename T>
static hipError_t launch_gemm_t(int epi, const GemmArgs& g, hipStream_t s) {
    const bool ok2 = (g.M % BM2 == 0) && (g.M >= 4 * BM2);
    const bool ok3 = ok2 && (g.N % BN3 == 0);
    const bool ok4 = ok3 && sizeof(T) == 2 && (g.K % 32 == 0);
    const bool ok5 = ok2 && sizeof(T) == 2 && (g.K % 32 == 0);
    int variant = g_variant;
    // measured on MI355X (tools/kbench.py): v3 0.825 ms, v1/v2 0.975 ms per layer of 8 images; with fewer than ~200 big tiles
    // (single-image calls) the 128x128 kernel fills the 256 CUs better
    if (variant == 0) variant = (ok3 && (int64_t)(g.M / BM2) * (g.N / BN3) >= 200) ? 3 : 1;
    if (variant == 5 && !ok5) variant = ok3 ? 3 : 1;
    if (variant == 4 && !ok4) variant = ok3 ? 3 : 1;
    if (variant == 3 && !ok3) variant = 1;
    if (variant == 2 && !ok2) variant = 1;
    const int ntiles = (variant == 3 || variant == 4) ? (g.M / BM2) * (g.N / BN3)
                       : (variant == 2 || variant == 5) ? (g.M / BM2) * (g.N / BN) : (g.M / BM) * (g.N / BN);
    dim3 grid(ntiles), block((variant == 1 || variant == 5) ? 256 : 512);
#define RZ_CASE(E) \
    case E: if (variant == 5) { if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((gemm_kernel_v5<T, E>), grid, block, 0, s, g); } \
            else if (variant == 4) { if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((gemm_kernel_v4<T, E>), grid, block, 0, s, g); } \
            else if (variant == 3) hipLaunchKernelGGL((gemm_kernel_v3<T, E>), grid, block, 0, s, g); \
            else if (variant == 2) hipLaunchKernelGGL((gemm_kernel_v2<T, E>), grid, block, 0, s, g); \
            else hipLaunchKernelGGL((gemm_kernel<T, E>), grid, block, 0, s, g); break;
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

hipError_t launch_gemm(int dtype, int epi, const GemmArgs& g_in, hipStream_t s) {
    GemmArgs g = g_in;
    g.debug_flags = g_debug_flags;
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
