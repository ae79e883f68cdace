// radzero_hip — pieces shared by the GEMM translation units (gemm.hip, gemm7.hip): tile rasterisation and the fused
// epilogues (reference call sites: see gemm.hip).
#pragma once
#include <type_traits>
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

// the same with GROUP_M chosen at run time (gemm8.hip's tile-walk experiment, GemmArgs::raster >= 100)
__device__ __forceinline__ void tile_coords_rt(int group_m, int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int per_group = group_m * tiles_n;
    const int grp = id / per_group;
    const int first_m = grp * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int r = id - grp * per_group;
    tm = first_m + r % gsz;
    tn = r / gsz;
}

// Read-modify-write epilogues on the fp32 residual stream (EPI_RESID_SCALE, EPI_RESID_ADD) and the patch-embedding table
// add (EPI_PATCH) for one wave's 64x64 block.  These epilogues are LATENCY bound if written as "load, wait, compute,
// store" per MFMA tile (what the generic loop below compiles to: `if (g.bias)` per tile + s_waitcnt vmcnt(0) before every
// use => ONE or two 1 KB loads in flight per wave, 32 dependent round trips to HBM per 128x64 wave tile: measured 33 us
// of epilogue behind a 17 us K loop at K = 768, i.e. 4 MB in flight chip-wide / ~1 us = 4 TB/s).  Here the bias / scale
// vectors are loaded once and the residual loads are issued in batches of 8 before the first of a batch is consumed:
// 8 KB in flight per wave, 64 KB per CU, 16 MB chip-wide.
template <typename T, int EPI>
__device__ __forceinline__ void gemm_epilogue_rmw(const GemmArgs& g, const f32x4 (&acc)[4][4], int mw, int nw, int l15, int lg) {
    static_assert(EPI == EPI_RESID_SCALE || EPI == EPI_RESID_ADD || EPI == EPI_PATCH, "fp32 read-modify-write epilogues");
    f32x4 b4[4], s4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = nw + j * 16 + 4 * lg;
        b4[j] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == EPI_RESID_SCALE) s4[j] = *reinterpret_cast<const f32x4*>(g.scale + n);
    }
    // two batches of 8 loads (two 16-row blocks x four column blocks): 32 VGPRs of staging beside the 128 accumulators
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {
        f32x4 hv[2][4];
        float* dst[2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int m = mw + (ih * 2 + ii) * 16 + l15;
            const float* src;
            if constexpr (EPI == EPI_PATCH) {
                const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                src = g.scale + (int64_t)tok * g.N + nw + 4 * lg;
                dst[ii] = reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + nw + 4 * lg;
            } else {
                src = g.resid + (int64_t)m * g.ldr + nw + 4 * lg;
                dst[ii] = (EPI == EPI_RESID_SCALE) ? g.resid + (int64_t)m * g.ldr + nw + 4 * lg
                                                   : reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + nw + 4 * lg;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[ii][j] = *reinterpret_cast<const f32x4*>(src + j * 16);
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[ih * 2 + ii][j] + b4[j];
                if constexpr (EPI == EPI_RESID_SCALE) v = hv[ii][j] + s4[j] * v;      // h += lambda * (acc + bias)
                else v = v + hv[ii][j];                                              // acc + bias + resid | acc + table
                *reinterpret_cast<f32x4*>(dst[ii] + j * 16) = v;
            }
    }
}

// Fused-LayerNorm epilogues (gemm8.hip "Fused LayerNorm") for one wave's 64x64 block, direct stores: the 128x128 kernel's
// versions.  Element for element the arithmetic is the persistent kernel's (v8_epilogue16 / v8_epilogue_resid_ln), so
// which kernel a batch size selects does not change a single bit of the result.
template <typename T, int EPI>
__device__ __forceinline__ void gemm_epilogue_ln(const GemmArgs& g, const f32x4 (&acc)[4][4], int mw, int nw, int l15, int lg) {
    typedef typename Traits<T>::vec4 vec4_t;
    if constexpr (EPI == EPI_RESID_SCALE_LN || EPI == EPI_PATCH_LN) {
        constexpr bool PATCH = (EPI == EPI_PATCH_LN);      // v = acc + table[token][n] -> g.out, centring constant 0 (v8_epilogue_resid_ln<T, true>)
        f32x4 b4[4], s4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nw + j * 16 + 4 * lg;
            b4[j] = (!PATCH && g.bias) ? *reinterpret_cast<const f32x4*>(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
            s4[j] = PATCH ? (f32x4){0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(g.scale + n);
        }
        const int slice = nw >> 6;
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            f32x4 hv[2][4];
            float* dst[2];
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int m = mw + (ih * 2 + ii) * 16 + l15;
                const float* src;
                if constexpr (PATCH) {
                    src = g.scale + (int64_t)(m % g.rows_per_image) * g.N + nw + 4 * lg;
                    dst[ii] = reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + nw + 4 * lg;
                } else {
                    dst[ii] = g.resid + (int64_t)m * g.ldr + nw + 4 * lg;
                    src = dst[ii];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[ii][j] = *reinterpret_cast<const f32x4*>(src + j * 16);
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = ih * 2 + ii;
                const int m = mw + i * 16 + l15;
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (PATCH) hv[ii][j] = hv[ii][j] + acc[i][j];
                    else hv[ii][j] = hv[ii][j] + s4[j] * (acc[i][j] + b4[j]);
                    *reinterpret_cast<f32x4*>(dst[ii] + j * 16) = hv[ii][j];
                    sum += (hv[ii][j][0] + hv[ii][j][1]) + (hv[ii][j][2] + hv[ii][j][3]);
                }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float mean = sum * (1.0f / 64.0f);
                float m2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 d = hv[ii][j] - mean;
                    m2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                }
                m2 += __shfl_xor(m2, 16, 64);
                m2 += __shfl_xor(m2, 32, 64);
                if (lg == 0) *reinterpret_cast<f32x2*>(g.ln_part + ((int64_t)m * 12 + slice) * 2) = (f32x2){mean, m2};
                const float cm = PATCH ? 0.f : g.ln_mu[m];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 gv = (hv[ii][j] - cm) * *reinterpret_cast<const f32x4*>(g.ln_gamma + nw + j * 16 + 4 * lg);
                    *reinterpret_cast<vec4_t*>(reinterpret_cast<T*>(g.ln_hb) + (int64_t)m * g.N + nw + j * 16 + 4 * lg) =
                        pack4<T>(gv[0], gv[1], gv[2], gv[3]);
                }
            }
        }
    } else if constexpr (EPI == EPI_VT_LN) {
        // lane owns column n, rows m..m+3 (4 consecutive tokens of one image)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mw + i * 16 + 4 * lg;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(g.ln_stat + 2 * (int64_t)m);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(g.ln_stat + 2 * (int64_t)m + 4);
            const f32x4 mu = (f32x4){p0[0], p0[2], p1[0], p1[2]}, rs = (f32x4){p0[1], p0[3], p1[1], p1[3]};
            const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nw + j * 16 + l15;
                const f32x4 v = (acc[i][j] - mu * g.scale[n]) * rs + g.bias[n];
                T* o = reinterpret_cast<T*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * 64 + (n & 63)) * g.rows_per_image + tok;
                *reinterpret_cast<vec4_t*>(o) = pack4<T>(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
        static_assert(EPI == EPI_HEADS_LN || EPI == EPI_GELU_LN, "fused-LayerNorm consumer epilogues");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mw + i * 16 + l15;
            const f32x2 st = *reinterpret_cast<const f32x2*>(g.ln_stat + 2 * (int64_t)m);
            const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nw + j * 16 + 4 * lg;
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(g.scale + n), c2 = *reinterpret_cast<const f32x4*>(g.bias + n);
                f32x4 v = (acc[i][j] - c1 * st[0]) * st[1] + c2;
                T* o;
                if constexpr (EPI == EPI_GELU_LN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_for<T>(v[e]);
                    o = reinterpret_cast<T*>(g.out) + (int64_t)m * g.ldo + n;
                } else {
                    o = reinterpret_cast<T*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * g.rows_per_image + tok) * 64 + (n & 63);
                }
                *reinterpret_cast<vec4_t*>(o) = pack4<T>(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// Fused epilogue for one wave's 64x64 accumulator block (4x4 MFMA tiles); (mw, nw) = block origin.
template <typename T, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x4 (&acc)[4][4], int mw, int nw, int l15, int lg) {
    constexpr bool SWAP = (EPI != EPI_VT && EPI != EPI_VT_LN);
    if constexpr (EPI == EPI_RESID_SCALE || EPI == EPI_RESID_ADD || EPI == EPI_PATCH) {
        gemm_epilogue_rmw<T, EPI>(g, acc, mw, nw, l15, lg);
        return;
    }
    if constexpr (EPI == EPI_RESID_SCALE_LN || EPI == EPI_PATCH_LN || EPI == EPI_HEADS_LN || EPI == EPI_VT_LN || EPI == EPI_GELU_LN) {
        gemm_epilogue_ln<T, EPI>(g, acc, mw, nw, l15, lg);
        return;
    }
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
                if constexpr (std::is_same<T, split_mx>::value) {
                    // fp32 mode, MX form: exact-erf GELU -> the next GEMM's A operand [hi f16 x N | per 64 columns: lo8 x 64, hi8 x 64] (rz_common.h)
                    static_assert(EPI == EPI_GELU, "MX-form output: the fc1 epilogue");
                    f16x4 hi;
                    uint32_t lo8, hi8;
                    split4_mx((f32x4){gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])}, hi, lo8, hi8, MX_A_HI_SCALE, MX_A_LO_SCALE, g.ovf_flag);
                    char* row = reinterpret_cast<char*>(g.out) + (int64_t)m * 4 * g.ldo;
                    *reinterpret_cast<f16x4*>(row + 2 * n) = hi;
                    char* pr = row + mx_pair_off((int)g.ldo, n);
                    *reinterpret_cast<uint32_t*>(pr) = lo8;
                    *reinterpret_cast<uint32_t*>(pr + 64) = hi8;
                } else if constexpr (std::is_same<T, split_f16>::value) {
                    // fp32 mode, hi/lo-split outputs (rz_common.h split4): EPI_GELU -> the next GEMM's A operand [M][3N] = [hi | lo | hi]
                    // (exact-erf GELU as in every fp32 epilogue); EPI_HEADS -> two planes `plane_off` elements apart
                    f16x4 hi, lo;
                    if constexpr (EPI == EPI_GELU) {
                        split4((f32x4){gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])}, hi, lo, g.ovf_flag);
                        f16_t* o = reinterpret_cast<f16_t*>(g.out) + (int64_t)m * 3 * g.ldo + n;
                        *reinterpret_cast<f16x4*>(o) = hi;
                        *reinterpret_cast<f16x4*>(o + g.ldo) = lo;
                        *reinterpret_cast<f16x4*>(o + 2 * g.ldo) = hi;
                    } else {
                        static_assert(EPI == EPI_HEADS, "split outputs: EPI_GELU, EPI_HEADS, EPI_VT");
                        split4(v, hi, lo, g.ovf_flag);
                        const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                        f16_t* o = reinterpret_cast<f16_t*>(g.out) +
                                   (((int64_t)b * g.heads_total + (n >> 6)) * g.rows_per_image + tok) * 64 + (n & 63);
                        *reinterpret_cast<f16x4*>(o) = hi;
                        *reinterpret_cast<f16x4*>(o + g.plane_off) = lo;
                    }
                } else if constexpr (EPI == EPI_STORE) {
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
                const int64_t idx = (((int64_t)b * g.heads_total + (n >> 6)) * 64 + (n & 63)) * g.rows_per_image + tok;
                if constexpr (std::is_same<T, split_f16>::value) {
                    f16x4 hi, lo;
                    split4((f32x4){a[0] + bv, a[1] + bv, a[2] + bv, a[3] + bv}, hi, lo, g.ovf_flag);
                    f16_t* o = reinterpret_cast<f16_t*>(g.out) + idx;
                    *reinterpret_cast<f16x4*>(o) = hi;
                    *reinterpret_cast<f16x4*>(o + g.plane_off) = lo;
                } else if constexpr (std::is_same<T, split_mxa>::value) {
                    // V^T for the MX attention (gemm_epilogue_tile_split FORM 2): hi f16 plane + pair plane `plane_off` elements behind it, 128 bytes
                    // [hi8 x 64 | lo8 x 64] per (feature, 64-token block), the block's tokens in the order the attention's score accumulators hand keys to a lane
                    f16x4 hi;
                    uint32_t lo8, hi8;
                    split4_mx((f32x4){a[0] + bv, a[1] + bv, a[2] + bv, a[3] + bv}, hi, lo8, hi8, MX_A_HI_SCALE, MX_A_LO_SCALE, g.ovf_flag);
                    f16_t* o = reinterpret_cast<f16_t*>(g.out) + idx;
                    *reinterpret_cast<f16x4*>(o) = hi;
                    const int kq = tok & 63, pos = 16 * ((kq & 31) >> 3) + 8 * (kq >> 5) + (kq & 7);
                    char* pr = reinterpret_cast<char*>(reinterpret_cast<f16_t*>(g.out) + g.plane_off + (idx - kq)) + pos;
                    *reinterpret_cast<uint32_t*>(pr) = hi8;
                    *reinterpret_cast<uint32_t*>(pr + 64) = lo8;
                } else {
                    T* o = reinterpret_cast<T*>(g.out) + idx;
                    *reinterpret_cast<typename Traits<T>::vec4*>(o) = pack4<T>(a[0] + bv, a[1] + bv, a[2] + bv, a[3] + bv);
                }
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

// ---------------------------------------------------------------------------------------------------
// Whole-tile 16-bit epilogue for the 8-wave 256x256 kernels (wave (wr, wc) owns rows wr*128.., columns wc*64..; acc[a]
// = its 64x64 block a).  Every lane converts its accumulators (bias, GELU) to T and drops them as 8-byte pieces into ONE
// 256-row x 512-byte image of the output tile in LDS (the operand buffers are free by then): 128 KB of LDS writes +
// 128 KB of reads instead of the 4 x 128 KB of the per-wave fp32 staging above, and the global stores become
// 2 rows x 512 contiguous bytes per wave instruction instead of 8 rows x 128.  Image row r holds output row m0+r
// (EPI_VT: output FEATURE n0+r, tokens along the row); 16-byte chunk c of row r sits at chunk position c ^ (r & 15), so the
// 16 rows x 4 pieces of a write instruction spread over all banks and a row is read back as 512 contiguous bytes.
// ---------------------------------------------------------------------------------------------------
template <typename T, int EPI>
__device__ __forceinline__ void gemm_epilogue_tile16(const GemmArgs& g, const f32x4 (&acc)[2][4][4], char* lds, int m0, int n0,
                                                     int wr, int wc, int lane, int tid) {
    static_assert(sizeof(T) == 2 && (EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_HEADS || EPI == EPI_VT), "16-bit row-major outputs only");
    constexpr bool SWAP = (EPI != EPI_VT);
    typedef typename Traits<T>::vec4 vec4_t;
    const int l15 = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            float bv = 0.f;
            if constexpr (SWAP) { if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + n0 + wc * 64 + j * 16 + 4 * lg); }
            else { if (g.bias) bv = g.bias[n0 + wc * 64 + j * 16 + l15]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = acc[a][i][j];
                int row, c16;
                if constexpr (SWAP) {
                    v += b4;
                    row = wr * 128 + a * 64 + i * 16 + l15;                 // output row
                    c16 = wc * 8 + j * 2 + (lg >> 1);                       // 8 columns per 16-byte chunk
                } else {
                    v += bv;
                    row = wc * 64 + j * 16 + l15;                           // output feature
                    c16 = wr * 16 + a * 8 + i * 2 + (lg >> 1);              // 8 tokens per chunk
                }
                if constexpr (EPI == EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_for<T>(v[e]);
                }
                *reinterpret_cast<vec4_t*>(lds + row * 512 + ((c16 ^ (row & 15)) << 4) + (lg & 1) * 8) = pack4<T>(v[0], v[1], v[2], v[3]);
            }
        }
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int q = it * 512 + tid;
        const int row = q >> 5, c16 = q & 31;
        const typename Traits<T>::frag v = *reinterpret_cast<const typename Traits<T>::frag*>(lds + row * 512 + ((c16 ^ (row & 15)) << 4));
        T* o;
        if constexpr (EPI == EPI_VT) {
            const int n = n0 + row, m = m0 + c16 * 8;
            const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
            o = reinterpret_cast<T*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * 64 + (n & 63)) * g.rows_per_image + tok;
        } else if constexpr (EPI == EPI_HEADS) {
            const int m = m0 + row, n = n0 + c16 * 8;
            const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
            o = reinterpret_cast<T*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * g.rows_per_image + tok) * 64 + (n & 63);
        } else {
            o = reinterpret_cast<T*>(g.out) + (int64_t)(m0 + row) * g.ldo + n0 + c16 * 8;
        }
        *reinterpret_cast<typename Traits<T>::frag*>(o) = v;
    }
}


// ---------------------------------------------------------------------------------------------------
// Whole-tile epilogue of the fp32 mode's SPLIT outputs for the 8-wave 256x256 kernels (round 4).  The direct form (gemm_epilogue above)
// writes 8- and 4-byte pieces straight from the accumulator layout — 32-64 contiguous bytes per row per instruction — and cost 40-60 us per
// tile against a 35-50 us K loop (profiles/r04: the fp32 mode's q|k, V^T and fc1 GEMMs were epilogue-bound).  Here the final fp32 values
// (bias, exact-erf GELU) are formed once in registers and leave plane by plane through the 256-row x 512-byte LDS image of
// gemm_epilogue_tile16 — 2 rows x 512 contiguous bytes per wave instruction:
//   FORM 0 (three f16 planes): hi = f16(v), lo = f16(v - hi);  EPI_HEADS / EPI_VT -> planes `plane_off` elements apart;
//                              EPI_GELU -> the next GEMM's A operand [M][3 ldo] = [hi | lo | hi]
//   FORM 1 (MX, rz_common.h):  EPI_GELU -> [hi f16 x ldo | per 64 columns: lo8 x 64, hi8 x 64]: the tile's 256 columns are 512 bytes of
//                              the hi plane and 512 bytes (4 groups) of the pair plane
// ---------------------------------------------------------------------------------------------------
//   FORM 2 (MX attention operands, attention.hip "MXA"): EPI_HEADS / EPI_VT -> hi f16 plane + a pair plane `plane_off` elements behind it with
//                              128 bytes of e4m3 per 64 elements: q heads [lo8 | hi8], k heads and V^T [hi8 | lo8]; V^T's 64 tokens of a
//                              block sit at position 16 g + j = token 8 g + j (j < 8) | 32 + 8 g + j - 8 (the order the attention's
//                              score accumulators hand keys to a lane)
template <int EPI, int FORM>
__device__ __forceinline__ void gemm_epilogue_tile_split(const GemmArgs& g, f32x4 (&acc)[2][4][4], char* lds, int m0, int n0,
                                                         int wr, int wc, int lane, int tid) {
    static_assert(EPI == EPI_GELU || ((EPI == EPI_HEADS || EPI == EPI_VT) && (FORM == 0 || FORM == 2)), "split outputs: q|k, V^T (planes), fc1 (either form)");
    static_assert(!(EPI == EPI_GELU && FORM == 2), "FORM 2 is the attention operands' form");
    constexpr bool SWAP = (EPI != EPI_VT);
    const int l15 = lane & 15, lg = lane >> 4;
    // final values, in place
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            float bv = 0.f;
            if constexpr (SWAP) { if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + n0 + wc * 64 + j * 16 + 4 * lg); }
            else { if (g.bias) bv = g.bias[n0 + wc * 64 + j * 16 + l15]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = acc[a][i][j];
                if constexpr (SWAP) v += b4; else v += bv;
                if constexpr (EPI == EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                }
                acc[a][i][j] = v;
            }
        }
    constexpr int NPASS = (FORM == 0 ? 2 : 2);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass) __syncthreads();              // the previous plane has been read out of the image
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = acc[a][i][j];
                    int row, c16;
                    if constexpr (SWAP) { row = wr * 128 + a * 64 + i * 16 + l15; c16 = wc * 8 + j * 2 + (lg >> 1); }
                    else { row = wc * 64 + j * 16 + l15; c16 = wr * 16 + a * 8 + i * 2 + (lg >> 1); }
                    const f16x4 hi = pack4<f16_t>(v[0], v[1], v[2], v[3]);
                    if (FORM == 2 && pass == 1) {
                        uint32_t lo8, hi8;
                        pair4_mx(v, hi, lo8, hi8, MX_A_HI_SCALE, MX_A_LO_SCALE);
                        int c_first, byte;            // 16-byte chunk (of the row's 32) and byte in it of the FIRST 4-byte piece; the second is 4 chunks on
                        bool hi_first;
                        if constexpr (SWAP) {         // q|k: the wave's 64 columns are one head = one 128-byte pair row
                            c_first = wc * 8 + j; byte = 4 * lg;
                            hi_first = (n0 >> 6) + wc >= (g.heads_total >> 1);       // k heads: [hi8 | lo8]; q heads: [lo8 | hi8]
                        } else {                      // V^T: tokens along the row; 64-token block (wr, a), token i * 16 + 4 lg + r inside it
                            const int kq = i * 16 + 4 * lg, pos = 16 * ((kq & 31) >> 3) + 8 * (kq >> 5) + (kq & 7);
                            c_first = (wr * 2 + a) * 8 + (pos >> 4); byte = pos & 15;
                            hi_first = true;
                        }
                        *reinterpret_cast<uint32_t*>(lds + row * 512 + ((c_first ^ (row & 15)) << 4) + byte) = hi_first ? hi8 : lo8;
                        *reinterpret_cast<uint32_t*>(lds + row * 512 + (((c_first + 4) ^ (row & 15)) << 4) + byte) = hi_first ? lo8 : hi8;
                    } else if (FORM == 1 && pass == 1) {
                        // pair plane: 4 bytes lo8 at byte 128 wc + 16 j + 4 lg of the row's 512, 4 bytes hi8 64 bytes further
                        uint32_t lo8, hi8;
                        pair4_mx(v, hi, lo8, hi8, MX_A_HI_SCALE, MX_A_LO_SCALE);
                        const int cl = wc * 8 + j, ch = cl + 4;
                        *reinterpret_cast<uint32_t*>(lds + row * 512 + ((cl ^ (row & 15)) << 4) + 4 * lg) = lo8;
                        *reinterpret_cast<uint32_t*>(lds + row * 512 + ((ch ^ (row & 15)) << 4) + 4 * lg) = hi8;
                    } else {
                        f16x4 o = hi;
                        if (pass == 0) { if (FORM == 0) flag_f16_range(v, g.ovf_flag); else flag_mx_range(v, MX_A_HI_SCALE, g.ovf_flag); }
                        else o = pack4<f16_t>(v[0] - (float)hi[0], v[1] - (float)hi[1], v[2] - (float)hi[2], v[3] - (float)hi[3]);
                        *reinterpret_cast<f16x4*>(lds + row * 512 + ((c16 ^ (row & 15)) << 4) + (lg & 1) * 8) = o;
                    }
                }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int q = it * 512 + tid;
            const int row = q >> 5, c16 = q & 31;
            const f16x8 v = *reinterpret_cast<const f16x8*>(lds + row * 512 + ((c16 ^ (row & 15)) << 4));
            if constexpr (EPI == EPI_GELU) {
                const int m = m0 + row;
                if constexpr (FORM == 0) {
                    f16_t* o = reinterpret_cast<f16_t*>(g.out) + (int64_t)m * 3 * g.ldo + n0 + c16 * 8;
                    if (pass == 0) { *reinterpret_cast<f16x8*>(o) = v; *reinterpret_cast<f16x8*>(o + 2 * g.ldo) = v; }
                    else *reinterpret_cast<f16x8*>(o + g.ldo) = v;
                } else {
                    char* rowp = reinterpret_cast<char*>(g.out) + (int64_t)m * 4 * g.ldo;
                    if (pass == 0) *reinterpret_cast<f16x8*>(rowp + 2 * (n0 + c16 * 8)) = v;
                    else *reinterpret_cast<f16x8*>(rowp + 2 * g.ldo + (n0 >> 6) * 128 + c16 * 16) = v;
                }
            } else {
                f16_t* o;
                if constexpr (EPI == EPI_VT) {
                    const int n = n0 + row, m = m0 + c16 * 8;
                    const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                    o = reinterpret_cast<f16_t*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * 64 + (n & 63)) * g.rows_per_image + tok;
                } else {
                    const int m = m0 + row, n = n0 + c16 * 8;
                    const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                    o = reinterpret_cast<f16_t*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * g.rows_per_image + tok) * 64 + (n & 63);
                }
                *reinterpret_cast<f16x8*>(pass == 0 ? o : o + g.plane_off) = v;
            }
        }
    }
}

}  // namespace rz
