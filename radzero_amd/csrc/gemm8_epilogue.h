// radzero_hip — wave-private epilogues of the persistent 256x256 GEMM kernels (gemm8.hip: 8 waves x 128x64; gemm10.hip: 4 waves x two
// 128x64 halves).  A "wave block" here is 128 rows x 64 columns held as acc[a][i][j] (a = 64-row half, i / j = 16-row / 16-column tile)
// with a 4 KB LDS region of its own: no workgroup barrier, stores left in flight under the next tile's K loop.
#pragma once
#include "gemm_common.h"

namespace rz {

// ---------------------------------------------------------------------------------------------------
// Wave-private epilogue for the 16-bit outputs.  The wave owns 128 x 64 outputs (acc[a] = its 64 x 64 block a).  Eight
// groups of 16 "outer" rows x 64 "inner" elements (SWAP: outer = output row m, inner = column n, group = (a, i);
// EPI_VT: outer = feature n, inner = token m, group = (j, a)) go through a 2 KB LDS image each: four 8-byte pieces per
// lane in, two 16-byte chunks per lane out, stored as 8 rows x 128 contiguous bytes per wave instruction.  Same wave,
// in-order LDS queue: no barrier, and the two halves of the 4 KB region alternate so a group's writes never wait for
// the previous group's reads.  16 global stores per lane.
// ---------------------------------------------------------------------------------------------------
template <typename T, int EPI, bool LNF = false>
__device__ __forceinline__ void v8_epilogue16(const GemmArgs& g, void* out, int heads_total, int n_rel0,
                                              const f32x4 (&acc)[2][4][4], char* wl, int mw, int nw, int lane) {
    static_assert(sizeof(T) == 2 && (EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_HEADS || EPI == EPI_VT), "16-bit outputs only");
    constexpr bool SWAP = (EPI != EPI_VT);
    typedef typename Traits<T>::vec4 vec4_t;
    typedef typename Traits<T>::frag frag_t;
    const int l15 = lane & 15, lg = lane >> 4;
    f32x4 b4[4], c4[4];      // per column block: bias (LNF: c2) and, LNF only, c1
    float bv[4], cv[4];
    // LNF: the wave's 128 (mean, rstd) pairs and its 64 c1 / c2 values were put into the upper half of its LDS region by
    // v8_prefetch_ln one tile ago (LDS-DMA, retired by the K loop's counted waits long before this point), so this epilogue
    // issues no global load at all and never has to drain the operand units that are in flight for the next tile.
    const float* lst = reinterpret_cast<const float*>(wl + 2048);          // [128][2]
    const float* lc1 = reinterpret_cast<const float*>(wl + 3072);          // [64]
    const float* lc2 = lc1 + 64;                                           // [64]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b4[j] = c4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bv[j] = cv[j] = 0.f;
        if constexpr (LNF) {
            if constexpr (SWAP) {
                b4[j] = *reinterpret_cast<const f32x4*>(lc2 + j * 16 + 4 * lg);
                c4[j] = *reinterpret_cast<const f32x4*>(lc1 + j * 16 + 4 * lg);
            } else {
                bv[j] = lc2[j * 16 + l15];
                cv[j] = lc1[j * 16 + l15];
            }
        } else if (g.bias) {
            if constexpr (SWAP) b4[j] = *reinterpret_cast<const f32x4*>(g.bias + nw + j * 16 + 4 * lg);
            else bv[j] = g.bias[nw + j * 16 + l15];
        }
    }
    // LNF: (mean, rstd) of the operand rows this lane's accumulators belong to
    f32x2 st_row[2][4];       // SWAP: row a*64 + i*16 + l15
    f32x4 st_mu[2][4], st_rs[2][4];   // VT: rows a*64 + i*16 + 4*lg + r
    if constexpr (LNF) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (SWAP) {
                    st_row[a][i] = *reinterpret_cast<const f32x2*>(lst + 2 * (a * 64 + i * 16 + l15));
                } else {
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(lst + 2 * (a * 64 + i * 16 + 4 * lg));
                    const f32x4 p1 = *reinterpret_cast<const f32x4*>(lst + 2 * (a * 64 + i * 16 + 4 * lg) + 4);
                    st_mu[a][i] = (f32x4){p0[0], p0[2], p1[0], p1[2]};
                    st_rs[a][i] = (f32x4){p0[1], p0[3], p1[1], p1[3]};
                }
            }
    }
    const unsigned wr_off = (unsigned)(l15 * 128 + (lg & 1) * 8);
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
        char* img = wl + (LNF ? 0 : (grp & 1) * 2048);        // LNF: one staging image, the other 2 KB hold the prefetched vectors
        const int a = SWAP ? (grp >> 2) : (grp & 1);
        const int x = SWAP ? (grp & 3) : (grp >> 1);          // SWAP: i (row block);  VT: j (feature block)
#pragma unroll
        for (int y = 0; y < 4; ++y) {                        // SWAP: j (column block); VT: i (token block)
            f32x4 v = SWAP ? acc[a][x][y] : acc[a][y][x];
            if constexpr (LNF) {
                if constexpr (SWAP) v = (v - c4[y] * st_row[a][x][0]) * st_row[a][x][1] + b4[y];
                else v = (v - st_mu[a][y] * cv[x]) * st_rs[a][y] + bv[x];
            } else {
                if constexpr (SWAP) v += b4[y]; else v += bv[x];
            }
            if constexpr (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_for<T>(v[e]);
            }
            const int c = y * 2 + (lg >> 1);                 // 16-byte chunk along inner
            *reinterpret_cast<vec4_t*>(img + wr_off + ((c ^ (l15 & 7)) << 4)) = pack4<T>(v[0], v[1], v[2], v[3]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * 64 + lane;
            const int row = q >> 3, c = q & 7;
            const frag_t v = *reinterpret_cast<const frag_t*>(img + row * 128 + ((c ^ (row & 7)) << 4));
            T* o;
            if constexpr (EPI == EPI_VT) {
                const int n = nw + x * 16 + row - n_rel0, m = mw + a * 64 + c * 8;
                const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                o = reinterpret_cast<T*>(out) + (((int64_t)b * heads_total + (n >> 6)) * 64 + (n & 63)) * g.rows_per_image + tok;
            } else if constexpr (EPI == EPI_HEADS) {
                const int m = mw + a * 64 + x * 16 + row, n = nw + c * 8 - n_rel0;
                const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                o = reinterpret_cast<T*>(out) + (((int64_t)b * heads_total + (n >> 6)) * g.rows_per_image + tok) * 64 + (n & 63);
            } else {
                o = reinterpret_cast<T*>(out) + (int64_t)(mw + a * 64 + x * 16 + row) * g.ldo + nw + c * 8;
            }
            *reinterpret_cast<frag_t*>(o) = v;
        }
        asm volatile("" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------------------
// Wave-private epilogues for the fp32 mode's SPLIT outputs (gemm_common.h gemm_epilogue_tile_split is the whole-tile version of the one-tile-
// per-workgroup kernel: same values, same bytes, same destinations).  Each of the eight 16 x 64 groups of v8_epilogue16 leaves as TWO planes
// through the wave's two 2 KB images — plane 0 = f16(v), plane 1 by FORM:
//   FORM 0  EPI_HEADS (q | k): lo = f16(v - hi), `plane_off` elements behind the hi plane
//   FORM 1  EPI_GELU (fc1 -> fc2's A operand in the MX form, rz_common.h): the group's 64 columns are ONE 128-byte pair block [lo8 x 64 | hi8 x 64]
//           at byte 2 ldo + (column / 64) * 128 of the output row
//   FORM 2  EPI_VT (V^T for the MX attention): per (feature, 64-token block) 128 bytes [hi8 x 64 | lo8 x 64], `plane_off` elements behind the hi plane,
//           the block's tokens in the order the attention's score accumulators hand keys to a lane (attention.hip "MXA")
// Values are formed once per group (bias, exact-erf GELU) and kept in registers for both planes.  32 global stores per lane.
// ---------------------------------------------------------------------------------------------------
template <int EPI, int FORM>
__device__ __forceinline__ void v8_epilogue_split(const GemmArgs& g, const f32x4 (&acc)[2][4][4], char* wl, int mw, int nw, int lane) {
    static_assert((EPI == EPI_HEADS && FORM == 0) || (EPI == EPI_GELU && FORM == 1) || (EPI == EPI_VT && FORM == 2), "the fp32 mode's default output forms");
    constexpr bool SWAP = (EPI != EPI_VT);
    const int l15 = lane & 15, lg = lane >> 4;
    f32x4 b4[4];
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bv[j] = 0.f;
        if (g.bias) {
            if constexpr (SWAP) b4[j] = *reinterpret_cast<const f32x4*>(g.bias + nw + j * 16 + 4 * lg);
            else bv[j] = g.bias[nw + j * 16 + l15];
        }
    }
    const unsigned wr_off = (unsigned)(l15 * 128 + (lg & 1) * 8);
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
        const int a = SWAP ? (grp >> 2) : (grp & 1);
        const int x = SWAP ? (grp & 3) : (grp >> 1);          // SWAP: i (row block);  VT: j (feature block)
        f32x4 v[4];
        f16x4 hi[4];
#pragma unroll
        for (int y = 0; y < 4; ++y) {                        // SWAP: j (column block); VT: i (token block)
            v[y] = SWAP ? acc[a][x][y] : acc[a][y][x];
            if constexpr (SWAP) v[y] += b4[y]; else v[y] += bv[x];
            if constexpr (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[y][e] = gelu_erf(v[y][e]);
            }
            if constexpr (FORM == 0) flag_f16_range(v[y], g.ovf_flag); else flag_mx_range(v[y], MX_A_HI_SCALE, g.ovf_flag);
            hi[y] = pack4<f16_t>(v[y][0], v[y][1], v[y][2], v[y][3]);
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            char* img = wl + pass * 2048;
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                if (pass == 0) {
                    const int c = y * 2 + (lg >> 1);             // 16-byte chunk along inner
                    *reinterpret_cast<f16x4*>(img + wr_off + ((c ^ (l15 & 7)) << 4)) = hi[y];
                } else if constexpr (FORM == 0) {
                    f16x4 d, lo;
                    split4(v[y], d, lo);
                    const int c = y * 2 + (lg >> 1);
                    *reinterpret_cast<f16x4*>(img + wr_off + ((c ^ (l15 & 7)) << 4)) = lo;
                } else {
                    uint32_t lo8, hi8;
                    pair4_mx(v[y], hi[y], lo8, hi8, MX_A_HI_SCALE, MX_A_LO_SCALE);
                    int c_lo, c_hi, byte;                        // 16-byte chunks of the row's 128 bytes and the byte inside them
                    if constexpr (FORM == 1) { c_lo = y; c_hi = y + 4; byte = 4 * lg; }                 // [lo8 x 64 | hi8 x 64], columns 16 y + 4 lg ..
                    else {                                                                              // [hi8 x 64 | lo8 x 64], tokens in the attention's order
                        const int kq = y * 16 + 4 * lg, pos = 16 * ((kq & 31) >> 3) + 8 * (kq >> 5) + (kq & 7);
                        c_hi = pos >> 4; c_lo = c_hi + 4; byte = pos & 15;
                    }
                    *reinterpret_cast<uint32_t*>(img + l15 * 128 + ((c_lo ^ (l15 & 7)) << 4) + byte) = lo8;
                    *reinterpret_cast<uint32_t*>(img + l15 * 128 + ((c_hi ^ (l15 & 7)) << 4) + byte) = hi8;
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int q = it * 64 + lane;
                const int row = q >> 3, c = q & 7;
                const f16x8 o8 = *reinterpret_cast<const f16x8*>(img + row * 128 + ((c ^ (row & 7)) << 4));
                if constexpr (EPI == EPI_GELU) {
                    char* rowp = reinterpret_cast<char*>(g.out) + (int64_t)(mw + a * 64 + x * 16 + row) * 4 * g.ldo;
                    if (pass == 0) *reinterpret_cast<f16x8*>(rowp + 2 * (nw + c * 8)) = o8;
                    else *reinterpret_cast<f16x8*>(rowp + 2 * g.ldo + (nw >> 6) * 128 + c * 16) = o8;
                } else {
                    f16_t* o;
                    if constexpr (EPI == EPI_VT) {
                        const int n = nw + x * 16 + row, m = mw + a * 64 + c * 8;
                        const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                        o = reinterpret_cast<f16_t*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * 64 + (n & 63)) * g.rows_per_image + tok;
                    } else {
                        const int m = mw + a * 64 + x * 16 + row, n = nw + c * 8;
                        const int b = m / g.rows_per_image, tok = m - b * g.rows_per_image;
                        o = reinterpret_cast<f16_t*>(g.out) + (((int64_t)b * g.heads_total + (n >> 6)) * g.rows_per_image + tok) * 64 + (n & 63);
                    }
                    *reinterpret_cast<f16x8*>(pass == 0 ? o : o + g.plane_off) = o8;
                }
            }
            asm volatile("" ::: "memory");
        }
    }
}

// EPI_RESID_SCALE_LN: resid += scale * (acc + bias) as gemm_epilogue_rmw does it, plus what the next LayerNorm needs of
// the new residual v: its copy in T (staged through the wave's LDS region like the 16-bit epilogue) and, per output row,
// (mean, M2) of this wave's 64 columns -> ln_part[m][3 * (n0 / 256) ... ], slice index = column / 64.
// PATCH (EPI_PATCH_LN): the same for the patch embedding: v = acc + table[token][n] (g.scale = table [rows_per_image][N]) -> g.out (fp32), centring
// constant 0 (there is no earlier mean; |mean| << sigma is this path's standing assumption), no bias / LayerScale.
template <typename T, bool PATCH = false>
__device__ __forceinline__ void v8_epilogue_resid_ln(const GemmArgs& g, const f32x4 (&acc)[2][4][4], char* wl, int mw, int nw, int lane) {
    typedef typename Traits<T>::vec4 vec4_t;
    typedef typename Traits<T>::frag frag_t;
    const int l15 = lane & 15, lg = lane >> 4;
    // The wave's 64 bias / LayerScale / next-LayerNorm-gain values live in the upper half of its 4 KB LDS region for the duration of the
    // epilogue (lane l < 48 fetches one 16-byte piece): 48 VGPRs less than holding them in registers — what the second residual buffer
    // below needs — and no ordinary global load is left inside the pipelined part.
    float* vec = reinterpret_cast<float*>(wl + 2048);                 // [0,64) bias  [64,128) scale  [128,192) gamma
    if (lane < 48) {
        const int which = lane >> 4, c4 = (lane & 15) * 4;
        const float* src = which == 0 ? g.bias : which == 1 ? (PATCH ? nullptr : g.scale) : g.ln_gamma;
        const f32x4 v = src ? *reinterpret_cast<const f32x4*>(src + nw + c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(vec + which * 64 + c4) = v;
    }
    const unsigned wr_off = (unsigned)(l15 * 128 + (lg & 1) * 8);
    const int slice = nw >> 6;                               // 0..11 for N = 768
    // The residual rows arrive in four batches of 32 rows (8 x 16-byte loads + 2 centring constants per lane), and batch g+1 is
    // REQUESTED BEFORE batch g is consumed and stored: the epilogue used to walk four dependent HBM round trips per tile (9.4-10 us
    // per out-proj tile in the stamped build, a quarter of its K loop).  hipcc cannot be asked for that: with LDS-DMA pieces of the
    // next tile in flight it puts vmcnt(0) in front of the first use of any ordinary load (cdna_hip_programming.md §5, trap (b)),
    // which would also wait for the batch just requested.  So these loads are inline asm and the wait is counted by hand: when batch g
    // is needed, the only younger vector-memory operations of this wave are the 10 loads of batch g+1 (more, if the compiler put
    // something behind them, only makes the wait stricter).
    f32x4 hvb[2][2][4];
    float cmb[2][2];
    auto request = [&](int grp, f32x4 (&hv)[2][4], float (&cm)[2]) {
        const int a = grp >> 1, ih = grp & 1;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int m = mw + a * 64 + (ih * 2 + ii) * 16 + l15;
            const float* src = PATCH ? g.scale + (int64_t)(m % g.rows_per_image) * g.N + nw + 4 * lg : g.resid + (int64_t)m * g.ldr + nw + 4 * lg;
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(hv[ii][j]) : "v"(src + j * 16) : "memory");
            if constexpr (PATCH) cm[ii] = 0.f;
            else asm volatile("global_load_dword %0, %1, off" : "=v"(cm[ii]) : "v"(g.ln_mu + m) : "memory");
        }
    };
    request(0, hvb[0], cmb[0]);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        const int a = grp >> 1, ih = grp & 1;
        f32x4 (&hv)[2][4] = hvb[grp & 1];
        float (&cmv)[2] = cmb[grp & 1];
        float* dst[2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int64_t mrow = mw + a * 64 + (ih * 2 + ii) * 16 + l15;
            dst[ii] = (PATCH ? reinterpret_cast<float*>(g.out) + mrow * g.ldo : g.resid + mrow * g.ldr) + nw + 4 * lg;
        }
        if (grp < 3) {
            request(grp + 1, hvb[(grp + 1) & 1], cmb[(grp + 1) & 1]);
            if constexpr (PATCH)             // 8 loads per batch (no centring constants)
                asm volatile("s_waitcnt vmcnt(8)"
                             : "+v"(hv[0][0]), "+v"(hv[0][1]), "+v"(hv[0][2]), "+v"(hv[0][3]), "+v"(hv[1][0]), "+v"(hv[1][1]), "+v"(hv[1][2]), "+v"(hv[1][3]) :: "memory");
            else
            asm volatile("s_waitcnt vmcnt(10)"
                         : "+v"(hv[0][0]), "+v"(hv[0][1]), "+v"(hv[0][2]), "+v"(hv[0][3]), "+v"(hv[1][0]), "+v"(hv[1][1]), "+v"(hv[1][2]), "+v"(hv[1][3]),
                           "+v"(cmv[0]), "+v"(cmv[1]) :: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(hv[0][0]), "+v"(hv[0][1]), "+v"(hv[0][2]), "+v"(hv[0][3]), "+v"(hv[1][0]), "+v"(hv[1][1]), "+v"(hv[1][2]), "+v"(hv[1][3]),
                           "+v"(cmv[0]), "+v"(cmv[1]) :: "memory");
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = ih * 2 + ii;
            const int m = mw + a * 64 + i * 16 + l15;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(vec + j * 16 + 4 * lg), s4 = *reinterpret_cast<const f32x4*>(vec + 64 + j * 16 + 4 * lg);
                if constexpr (PATCH) hv[ii][j] = hv[ii][j] + acc[a][i][j];        // table + patch projection
                else hv[ii][j] = hv[ii][j] + s4 * (acc[a][i][j] + b4);            // the new residual
                *reinterpret_cast<f32x4*>(dst[ii] + j * 16) = hv[ii][j];
                sum += (hv[ii][j][0] + hv[ii][j][1]) + (hv[ii][j][2] + hv[ii][j][3]);
            }
            // the row's 64 values sit in the four lanes (l15, lg = 0..3): two-pass statistics across them
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
            // T copy, centred with the row's previous mean and scaled by the consuming LayerNorm's gain BEFORE rounding:
            // 16 rows x 64 columns through the wave's LDS image (ONE 2 KB image: same wave, in-order LDS queue), whole 128-byte row pieces out
            const float cm = cmv[ii];
            char* img = wl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = j * 2 + (lg >> 1);
                const f32x4 gv = (hv[ii][j] - cm) * *reinterpret_cast<const f32x4*>(vec + 128 + j * 16 + 4 * lg);
                *reinterpret_cast<vec4_t*>(img + wr_off + ((c ^ (l15 & 7)) << 4)) = pack4<T>(gv[0], gv[1], gv[2], gv[3]);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int q = it * 64 + lane;
                const int row = q >> 3, c = q & 7;
                const frag_t v = *reinterpret_cast<const frag_t*>(img + row * 128 + ((c ^ (row & 7)) << 4));
                *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(g.ln_hb) + (int64_t)(mw + a * 64 + i * 16 + row) * g.N + nw + c * 8) = v;
            }
            asm volatile("" ::: "memory");
        }
    }
}

// Fused-LayerNorm consumers: what the epilogue of the tile at (mw, nw) will need besides the accumulators — the 128 (mean, rstd)
// pairs of this wave's rows (1 KB, contiguous in ln_stat) and its 64 c1 and 64 c2 values — goes into the upper 2 KB of the
// wave's LDS region by two LDS-DMA instructions, issued one tile ahead (workgroup prologue / end of the previous epilogue).
__device__ __forceinline__ void v8_prefetch_ln(const GemmArgs& g, char* wl, int mw, int nw, int lane) {
    const char* s0 = reinterpret_cast<const char*>(g.ln_stat + 2 * (int64_t)mw) + lane * 16;
    // lanes 0-15: c1[nw ..], 16-31: c2[nw ..] (= g.scale / g.bias), lanes 32-63 repeat them into the 512 bytes behind
    const float* vec = (lane & 16) ? g.bias : g.scale;
    const char* s1 = reinterpret_cast<const char*>(vec + nw) + (lane & 15) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s0, (__attribute__((address_space(3))) void*)(wl + 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s1, (__attribute__((address_space(3))) void*)(wl + 3072), 16, 0, 0);
}

template <int EPI> struct V8Epi {
    // 16-byte stores a wave issues LAST in this epilogue (nothing but stores after them), halved: see header
    static constexpr int kExtra = (EPI == EPI_RESID_SCALE || EPI == EPI_RESID_SCALE_LN || EPI == EPI_PATCH_LN || EPI == EPI_RESID_ADD || EPI == EPI_PATCH || EPI == EPI_STORE_F32) ? 16 : 8;
};

template <typename T, int EPI, bool SWAP>
__device__ __forceinline__ void v8_epilogue(const GemmArgs& g, const f32x4 (&acc)[2][4][4], char* wl, int mw, int nw, int lane) {
    if constexpr (EPI == EPI_QKV || EPI == EPI_QKV_LN) {
        // merged q|k|v projection: columns [0, split_n) -> per-head q|k tensor, the rest -> transposed v tensor
        constexpr bool LNF = (EPI == EPI_QKV_LN);
        if constexpr (SWAP) v8_epilogue16<T, EPI_HEADS, LNF>(g, g.out, g.heads_total, 0, acc, wl, mw, nw, lane);
        else v8_epilogue16<T, EPI_VT, LNF>(g, g.out2, g.heads_total2, g.split_n, acc, wl, mw, nw, lane);
    } else if constexpr (EPI == EPI_GELU_LN) {
        v8_epilogue16<T, EPI_GELU, true>(g, g.out, g.heads_total, 0, acc, wl, mw, nw, lane);
    } else if constexpr (EPI == EPI_RESID_SCALE_LN) {
        v8_epilogue_resid_ln<T>(g, acc, wl, mw, nw, lane);
    } else if constexpr (EPI == EPI_PATCH_LN) {
        v8_epilogue_resid_ln<T, true>(g, acc, wl, mw, nw, lane);
    } else if constexpr (EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_HEADS || EPI == EPI_VT) {
        v8_epilogue16<T, EPI>(g, g.out, g.heads_total, 0, acc, wl, mw, nw, lane);
    } else {
        gemm_epilogue<T, EPI>(g, acc[0], mw, nw, lane & 15, lane >> 4);
        gemm_epilogue<T, EPI>(g, acc[1], mw + 64, nw, lane & 15, lane >> 4);
    }
}

}  // namespace rz
