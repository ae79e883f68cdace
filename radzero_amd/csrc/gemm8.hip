// radzero_hip — persistent 256x256x64 GEMM for 16-bit operands ("v8"): the staggered four-phase K loop of gemm7.hip,
// run as ONE workgroup per CU that walks a list of output tiles with the operand stream never interrupted.
//
// Why (profiles/r01/gemm_kslope_v3_v7.log): at K = 768 the one-tile-per-workgroup kernel pays ~12.6 us of fixed cost per
// 256x256 tile against a 17 us K loop — pipeline fill (64 KB of operands from beyond L2 before the first MFMA), the
// drain at the end, a workgroup-wide LDS-staged epilogue behind __syncthreads(), and the launch of the next workgroup.
// Here the LDS-DMA units of the NEXT tile's first two K tiles are issued during the last two K tiles of the current one
// (they are simply "tile T+1 / T+2" of the same pipeline), so the matrix pipe restarts one barrier after the epilogue's
// last store has been issued; the epilogue itself is wave-private (4 KB of LDS per wave outside the operand buffers, no
// workgroup barrier), and its stores stay in flight under the next tile's first phases.
//
// K loop: identical to gemm7.hip (same units, same phase order, same vmcnt accounting — read its header first).
// Differences:
//   * operand addresses are (wave-uniform base) + (32-bit per-lane offset): the per-lane part does not depend on the
//     tile, the base lives in SGPRs and is what changes at a tile seam;
//   * vmcnt counts EVERY vector-memory instruction of the wave in issue order, stores included.  The first K tile after
//     an epilogue therefore allows V8_EXTRA more outstanding operations in its first three waits (the epilogue's youngest
//     stores), and nothing else changes: every operand unit that those waits must retire is OLDER than the stores.
//     V8_EXTRA is deliberately half the number of 16-byte stores an epilogue issues last (8 of 16 / 16 of 32): a wait that
//     allows fewer outstanding operations than are really younger is always safe, one that allows more is a race.
//   * K / 64 must be even and >= 4 (buffer parity restarts at every tile): 640, 768, 3072 on this path.
// Fused LayerNorm (TF:dinov2/modeling_dinov2.py:348-353 norm1 / norm2 -> :199-213 q|k|v and :281-297 fc1): LN(x) W^T + b
//   = rstd_m * ( (x gamma) W^T - mu_m * c1 ) + c2   with  c1[n] = sum_k gamma[k] W[n][k],  c2 = W beta + b,
// so the GEMM that FOLLOWS a LayerNorm multiplies the un-normalised, gain-scaled residual with the block's ordinary weights
// (EPI_QKV_LN, EPI_GELU_LN: c1, c2 are packed by the host once, per-row (mu, rstd) come in ln_stat), and the GEMM that
// PRECEDES it (EPI_RESID_SCALE_LN: out-proj, fc2) writes, beside the fp32 residual stream, x gamma in the compute dtype (the
// gain goes on the activation BEFORE rounding, exactly where the un-fused path applies it: massive-activation channels with
// small gains are then rounded at their scaled size; it is also centred with the row's PREVIOUS mean, so that the
// consumer's mu_m is only the small change of the mean and acc - mu_m c1 cannot cancel) and per-row partial statistics of each 64-column
// slice (two-pass in registers: mean, then M2 about it; merged exactly by Chan's formula in rowops.hip::ln_finalize).  The 29
// stand-alone LayerNorm passes over the residual stream shrink to two row kernels per forward (before block 0 and the ViT's
// final LayerNorm).  Accuracy is that of the un-fused 16-bit path as long as |mu| is small against sigma for every token
// (the operand is rounded relative to |x gamma| instead of |(x - mu) gamma|): measured identical on the benign and the
// outlier-channel checkpoints (|mu|/sigma <= 0.12), see DESIGN.md.
// Tile order: XCD x owns the logical tile ids of xcd_remap's range x; its `grid/8` workgroups take consecutive ids
// round after round, so the 32 CUs of an XCD always work on one compact GROUP_M x n block of tiles (gemm_common.h).
#include <type_traits>

#include "gemm_common.h"

namespace rz {

constexpr int V8_BM = 256, V8_BN = 256;
constexpr int V8_STAGE = (V8_BM + V8_BN) * 128;     // 64 KB: A panel (256 rows x 128 B) then W panel
constexpr int V8_WAVE_LDS = 4096;                   // wave-private epilogue staging (two 2 KB halves)

template <typename T, bool SWAP>
__device__ __forceinline__ void v8_mma(f32x4& c, const typename Traits<T>::frag& a, const typename Traits<T>::frag& b) {
    if constexpr (SWAP) c = mma(b, a, c); else c = mma(a, b, c);
}

// `base` is wave-uniform (SGPR pair), `off` the lane's 32-bit byte offset: global_load_lds_dwordx4 v_off, s[base].
// The empty asm statements make both operands opaque at every use, so that loop strength reduction cannot turn
// base + zext(off) into loop-carried 64-bit per-lane pointers (16 more VGPRs and no SGPR-base addressing).
__device__ __forceinline__ void v8_glds(const char* base, unsigned off, char* lds_dst) {
    asm volatile("" : "+s"(base));
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void v8_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One K tile.  a1 / w1 (a2 / w2): wave-uniform source bases at the K offset of K tile T+1 (T+2), which may belong to the next
// output tile; the units of T+1 / T+2 are ALWAYS issued, so the counted waits hold everywhere — behind a workgroup's last
// output tile they re-fetch operands of that same tile into buffers nobody reads again (128 KB per workgroup per launch),
// and the kernel drains them before it ends.  extra (wave-uniform, run time): this is the first K tile after an epilogue
// (see header); one scalar branch per phase selects the wait.  ONE copy of this body exists in the kernel, inside one
// simple loop: with one inlined variant per case hipcc renamed the accumulators between the copies and spilled them.
template <typename T, bool SWAP, int EXTRA>
__device__ __forceinline__ void v8_tile(f32x4 (&acc)[2][4][4], char* cur, char* nxt, unsigned a_rd, unsigned b_rd,
                                        const char* a1, const char* w1, const char* a2, const char* w2,
                                        const unsigned (&a_off)[2], const unsigned (&w_off)[2], int64_t a_sub, int64_t w_sub,
                                        unsigned a_dst, unsigned w_dst, bool extra) {
    typedef typename Traits<T>::frag frag_t;
    frag_t fa[2][4], fb0[2][2], fb1[2][2];          // [k-half][fragment]
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        // ---- LOAD part
        if (u == 0 || u == 2) {
            const unsigned o = a_rd + (u == 2 ? 64 * 128 : 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[ks][i] = *reinterpret_cast<const frag_t*>(cur + ((o ^ (ks * 64)) + i * 2048));
        }
        if (u == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) fb0[ks][j] = *reinterpret_cast<const frag_t*>(cur + ((b_rd ^ (ks * 64)) + j * 2048));
        }
        if (u == 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) fb1[ks][j] = *reinterpret_cast<const frag_t*>(cur + (((b_rd + 32 * 128) ^ (ks * 64)) + j * 2048));
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (u == 0) v8_glds(w1 + w_sub, w_off[e], nxt + w_dst + 32 * 128 + e * 1024);     // U2(T+1)
            if (u == 1) v8_glds(a1 + a_sub, a_off[e], nxt + a_dst + 64 * 128 + e * 1024);     // U3(T+1)
            if (u == 2) v8_glds(a2, a_off[e], cur + a_dst + e * 1024);                        // U0(T+2)
            if (u == 3) v8_glds(w2, w_off[e], cur + w_dst + e * 1024);                        // U1(T+2)
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA part: quadrant (mi, ni) = (0,0) (0,1) (1,1) (1,0)
        const int mi = u >> 1, ni = (u == 1 || u == 2) ? 1 : 0;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    v8_mma<T, SWAP>(acc[mi][i][ni * 2 + j], fa[ks][i], ni ? fb1[ks][j] : fb0[ks][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // everything issued three or more phases ago must have landed: the three youngest units (6 instructions) may stay
        // in flight (+ EXTRA epilogue stores while they can still be among the youngest: phases 0-2 after an epilogue)
        if (u < 3 && extra) v8_wait_vm<6 + EXTRA>();
        else v8_wait_vm<6>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
}

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

// EPI_RESID_SCALE_LN: resid += scale * (acc + bias) as gemm_epilogue_rmw does it, plus what the next LayerNorm needs of
// the new residual v: its copy in T (staged through the wave's LDS region like the 16-bit epilogue) and, per output row,
// (mean, M2) of this wave's 64 columns -> ln_part[m][3 * (n0 / 256) ... ], slice index = column / 64.
template <typename T>
__device__ __forceinline__ void v8_epilogue_resid_ln(const GemmArgs& g, const f32x4 (&acc)[2][4][4], char* wl, int mw, int nw, int lane) {
    typedef typename Traits<T>::vec4 vec4_t;
    typedef typename Traits<T>::frag frag_t;
    const int l15 = lane & 15, lg = lane >> 4;
    f32x4 b4[4], s4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = nw + j * 16 + 4 * lg;
        b4[j] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        s4[j] = *reinterpret_cast<const f32x4*>(g.scale + n);
    }
    const unsigned wr_off = (unsigned)(l15 * 128 + (lg & 1) * 8);
    const int slice = nw >> 6;                               // 0..11 for N = 768
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {                      // 32 rows per batch: 8 loads in flight, as in gemm_epilogue_rmw
        const int a = grp >> 1, ih = grp & 1;
        f32x4 hv[2][4];
        float* dst[2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int m = mw + a * 64 + (ih * 2 + ii) * 16 + l15;
            dst[ii] = g.resid + (int64_t)m * g.ldr + nw + 4 * lg;
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[ii][j] = *reinterpret_cast<const f32x4*>(dst[ii] + j * 16);
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = ih * 2 + ii;
            const int m = mw + a * 64 + i * 16 + l15;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hv[ii][j] = hv[ii][j] + s4[j] * (acc[a][i][j] + b4[j]);           // the new residual
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
            // 16 rows x 64 columns through the wave's LDS image, whole 128-byte row pieces out
            const float cm = g.ln_mu[m];
            char* img = wl + (i & 1) * 2048;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = j * 2 + (lg >> 1);
                const f32x4 gv = (hv[ii][j] - cm) * *reinterpret_cast<const f32x4*>(g.ln_gamma + nw + j * 16 + 4 * lg);
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
    static constexpr int kExtra = (EPI == EPI_RESID_SCALE || EPI == EPI_RESID_SCALE_LN || EPI == EPI_RESID_ADD || EPI == EPI_PATCH || EPI == EPI_STORE_F32) ? 16 : 8;
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
    } else if constexpr (EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_HEADS || EPI == EPI_VT) {
        v8_epilogue16<T, EPI>(g, g.out, g.heads_total, 0, acc, wl, mw, nw, lane);
    } else {
        gemm_epilogue<T, EPI>(g, acc[0], mw, nw, lane & 15, lane >> 4);
        gemm_epilogue<T, EPI>(g, acc[1], mw + 64, nw, lane & 15, lane >> 4);
    }
}

// STAMP: diagnostic build (tools/kstamp8.py, compiled only with -DRZ_EXPERIMENTS; never launched by the model): every wave sums the 100 MHz real-time ticks it
// spends in K loops and in epilogues and stores them, with the absolute time of its first 24 epilogue starts, into the
// buffer passed as g.out2 — memory no other code of the kernel reads.
template <typename T, int EPI, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v8(GemmArgs g) {
    static_assert(sizeof(T) == 2, "v8 is for 16-bit operands");
    __shared__ __attribute__((aligned(1024))) char lds[2 * V8_STAGE + 8 * V8_WAVE_LDS];     // 160 KB: one workgroup per CU
    constexpr int EXTRA = V8Epi<(EPI == EPI_QKV || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN) ? EPI_HEADS : EPI>::kExtra;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- this workgroup's tile list
    const int tiles_n = g.N / V8_BN, tiles_m = g.M / V8_BM, ntiles = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, stride = gridDim.x >> 3;
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int cnt = tq + (xcd < tr ? 1 : 0);
    if (slot >= cnt) return;

    const int64_t lda_b = g.lda * 2, ldw_b = g.ldw * 2;
    const int nk = g.K / 64;

    // this wave's two DMA pieces (e = 0, 1) of each unit: 8 panel rows each (gemm7.hip):
    //   A units: rows wr*128 + sub*64 + ((wave&3)*2 + e)*8      W units: rows (wave>>1)*64 + sub*32 + ((wave&1)*2 + e)*8
    // lane l lands on row +(l>>3), chunk l&7, and fetches chunk (l&7) ^ swz_std(row) = (l&7) ^ ((4e + (l>>4)) & 7)
    const int a_row = wr * 128 + (wave & 3) * 16, w_row = (wave >> 1) * 64 + (wave & 1) * 16;
    unsigned a_off[2], w_off[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned sw = (unsigned)(((lane & 7) ^ ((4 * e + (lane >> 4)) & 7)) << 4);
        a_off[e] = (unsigned)((a_row + e * 8 + (lane >> 3)) * lda_b) + sw;
        w_off[e] = (unsigned)((w_row + e * 8 + (lane >> 3)) * ldw_b) + sw;
    }
    const int64_t a_sub = 64 * lda_b, w_sub = 32 * ldw_b;
    const unsigned a_dst = (unsigned)(a_row * 128), w_dst = (unsigned)(V8_BM * 128 + w_row * 128);
    const unsigned frd = (unsigned)(l15 * 128 + ((lg ^ ((l15 >> 1) & 7)) << 4));
    const unsigned a_rd = (unsigned)(wr * 128 * 128) + frd;
    const unsigned b_rd = (unsigned)(V8_BM * 128 + wc * 64 * 128) + frd;
    char* wl = lds + 2 * V8_STAGE + wave * V8_WAVE_LDS;

    auto tile_origin = [&](int idx, int& m0, int& n0) {
        int tm, tn;
        tile_coords<4>(lo + idx, tiles_m, tiles_n, tm, tn);
        m0 = tm * V8_BM;
        n0 = tn * V8_BN;
    };

    f32x4 acc[2][4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int idx = slot, m0, n0;
    tile_origin(idx, m0, n0);
    const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)m0 * lda_b;
    const char* Wb = reinterpret_cast<const char*>(g.W) + (int64_t)n0 * ldw_b;

    constexpr bool LN_CONSUMER = (EPI == EPI_QKV_LN || EPI == EPI_GELU_LN);
    if constexpr (LN_CONSUMER) v8_prefetch_ln(g, wl, m0 + wr * 128, n0 + wc * 64, lane);      // oldest in the queue: retired by the prologue wait
    // prologue (once per workgroup): K tile 0 complete in buffer 0; U0, U1 of K tile 1 on their way into buffer 1
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        v8_glds(Ab, a_off[e], lds + a_dst + e * 1024);
        v8_glds(Wb, w_off[e], lds + w_dst + e * 1024);
        v8_glds(Wb + w_sub, w_off[e], lds + w_dst + 32 * 128 + e * 1024);
        v8_glds(Ab + a_sub, a_off[e], lds + a_dst + 64 * 128 + e * 1024);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) v8_glds(Ab + 128, a_off[e], lds + V8_STAGE + a_dst + e * 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) v8_glds(Wb + 128, w_off[e], lds + V8_STAGE + w_dst + e * 1024);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    bool after_epilogue = false;
    unsigned long long st_k = 0, st_e = 0, st_n = 0, st_t = 0, st_c = 0, st_c0 = 0;
    unsigned long long* stamp = nullptr;
    if constexpr (STAMP) {
        stamp = reinterpret_cast<unsigned long long*>(g.out2) + ((size_t)blockIdx.x * 8 + wave) * 32;
        st_t = __builtin_amdgcn_s_memrealtime();
        st_c0 = __builtin_amdgcn_s_memtime();
        if (lane == 0) stamp[3] = st_t;
    }
    for (;;) {
        const bool has_next = idx + stride < cnt;
        int m1 = m0, n1 = n0;
        if (has_next) tile_origin(idx + stride, m1, n1);
        const char* An = reinterpret_cast<const char*>(g.A) + (int64_t)m1 * lda_b;
        const char* Wn = reinterpret_cast<const char*>(g.W) + (int64_t)n1 * ldw_b;
        const bool vt_tile = (EPI == EPI_VT) || ((EPI == EPI_QKV || EPI == EPI_QKV_LN) && n0 >= g.split_n);

        auto k_loop = [&](auto swap_c) {
            constexpr bool SWAP = decltype(swap_c)::value;
            // K tile 0 apart (its waits may allow the epilogue's stores: run-time flag), then the steady loop with the flag a
            // compile-time false: no branch between the MFMAs and the counted wait.  The merged q|k|v kernels hold both operand
            // orders of this loop; a second call site each made hipcc spill, so they keep the flag inside one loop.
            constexpr bool PEEL = !(EPI == EPI_QKV || EPI == EPI_QKV_LN);
            if constexpr (PEEL)
                v8_tile<T, SWAP, EXTRA>(acc, lds, lds + V8_STAGE, a_rd, b_rd, Ab + 128, Wb + 128, Ab + 256, Wb + 256, a_off, w_off, a_sub,
                                        w_sub, a_dst, w_dst, after_epilogue);
            for (int kt = PEEL ? 1 : 0; kt < nk; ++kt) {
                char* cur = lds + (kt & 1) * V8_STAGE;
                char* nxt = lds + ((kt + 1) & 1) * V8_STAGE;
                const bool in1 = kt + 1 < nk, in2 = kt + 2 < nk;
                const char* a1 = in1 ? Ab + (int64_t)(kt + 1) * 128 : An + (int64_t)(kt + 1 - nk) * 128;
                const char* w1 = in1 ? Wb + (int64_t)(kt + 1) * 128 : Wn + (int64_t)(kt + 1 - nk) * 128;
                const char* a2 = in2 ? Ab + (int64_t)(kt + 2) * 128 : An + (int64_t)(kt + 2 - nk) * 128;
                const char* w2 = in2 ? Wb + (int64_t)(kt + 2) * 128 : Wn + (int64_t)(kt + 2 - nk) * 128;
                v8_tile<T, SWAP, EXTRA>(acc, cur, nxt, a_rd, b_rd, a1, w1, a2, w2, a_off, w_off, a_sub, w_sub, a_dst, w_dst,
                                        PEEL ? false : (kt == 0 && after_epilogue));
            }
        };
        const int mw = m0 + wr * 128, nw = n0 + wc * 64;
        if (vt_tile) {
            if constexpr (EPI == EPI_VT || EPI == EPI_QKV || EPI == EPI_QKV_LN) {
                k_loop(std::integral_constant<bool, false>{});
                __builtin_amdgcn_sched_barrier(0);
                v8_epilogue<T, EPI, false>(g, acc, wl, mw, nw, lane);
            }
        } else {
            if constexpr (EPI != EPI_VT) {
                k_loop(std::integral_constant<bool, true>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (STAMP) {
                    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
                    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
                    st_c += c1 - st_c0;
                    st_k += t1 - st_t;
                    if (lane == 0 && st_n < 24) stamp[8 + st_n] = t1;
                    st_t = t1;
                    __builtin_amdgcn_sched_barrier(0);
                }
                v8_epilogue<T, EPI, true>(g, acc, wl, mw, nw, lane);
                if constexpr (STAMP) {
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
                    st_c0 = __builtin_amdgcn_s_memtime();
                    st_e += t2 - st_t;
                    st_t = t2;
                    st_n += 1;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        if constexpr (LN_CONSUMER) {       // the epilogue above has read its vectors: fetch the next tile's (youngest in the queue)
            v8_prefetch_ln(g, wl, m1 + wr * 128, n1 + wc * 64, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[a][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        idx += stride;
        m0 = m1; n0 = n1;
        Ab = An; Wb = Wn;
        after_epilogue = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the units issued past the last output tile (see v8_tile)
    if (wr == 0) __builtin_amdgcn_s_barrier();
    if constexpr (STAMP) {
        if (lane == 0) { stamp[0] = st_k; stamp[1] = st_e; stamp[2] = st_n; stamp[4] = __builtin_amdgcn_s_memrealtime(); stamp[5] = st_c; }
    }
}

static int v8_grid() {
    static int grid = 0;
    if (grid == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            cus = 256;
        grid = cus / 8 * 8;            // one 160 KB workgroup per CU; a multiple of 8 keeps `b & 7` = XCD label
    }
    return grid;
}

#ifdef RZ_EXPERIMENTS
static void* g_stamp_buf = nullptr;
void gemm_v8_set_stamp_buffer(void* p) { g_stamp_buf = p; }
#endif

template <typename T>
static hipError_t launch_v8_t(int epi, const GemmArgs& g_in, hipStream_t s) {
    dim3 grid(v8_grid()), block(512);
    GemmArgs g = g_in;
#ifdef RZ_EXPERIMENTS
    if (g_stamp_buf && (epi == EPI_HEADS || epi == EPI_GELU || epi == EPI_RESID_SCALE)) {     // diagnostic build, see STAMP above
        g.out2 = g_stamp_buf;
        if (epi == EPI_HEADS) hipLaunchKernelGGL((gemm_kernel_v8<T, EPI_HEADS, true>), grid, block, 0, s, g);
        else if (epi == EPI_GELU) hipLaunchKernelGGL((gemm_kernel_v8<T, EPI_GELU, true>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_kernel_v8<T, EPI_RESID_SCALE, true>), grid, block, 0, s, g);
        return hipGetLastError();
    }
#endif
#define RZ_CASE8(E) case E: hipLaunchKernelGGL((gemm_kernel_v8<T, E>), grid, block, 0, s, g); break;
    switch (epi) {
        RZ_CASE8(EPI_STORE)
        RZ_CASE8(EPI_GELU)
        RZ_CASE8(EPI_HEADS)
        RZ_CASE8(EPI_VT)
        RZ_CASE8(EPI_RESID_SCALE)
        RZ_CASE8(EPI_RESID_ADD)
        RZ_CASE8(EPI_PATCH)
        RZ_CASE8(EPI_STORE_F32)
        RZ_CASE8(EPI_QKV)
        RZ_CASE8(EPI_RESID_SCALE_LN)
        RZ_CASE8(EPI_QKV_LN)
        RZ_CASE8(EPI_GELU_LN)
        default: return hipErrorInvalidValue;
    }
#undef RZ_CASE8
    return hipGetLastError();
}

// shape contract: M % 256 == 0, N % 256 == 0, K % 128 == 0, K >= 256 (an even number >= 4 of 64-wide K tiles), 16-bit dtype,
// per-lane operand offsets < 4 GB; EPI_QKV additionally split_n % 256 == 0.
bool gemm_v8_ok(int dtype, int epi, const GemmArgs& g) {
    if (dtype == DT_F32 || g.M % V8_BM || g.N % V8_BN || g.K % 128 || g.K < 256) return false;
    if ((int64_t)256 * g.lda * 2 >= ((int64_t)1 << 32) || (int64_t)256 * g.ldw * 2 >= ((int64_t)1 << 32)) return false;
    if ((epi == EPI_QKV || epi == EPI_QKV_LN) && (g.split_n % V8_BN || g.split_n <= 0 || g.split_n >= g.N || !g.out2)) return false;
    if ((epi == EPI_QKV_LN || epi == EPI_GELU_LN) && (!g.ln_stat || !g.scale || !g.bias)) return false;
    if (epi == EPI_RESID_SCALE_LN && (g.N != 768 || !g.ln_part || !g.ln_hb || !g.ln_gamma || !g.ln_mu || !g.scale || !g.resid)) return false;
    return true;
}

hipError_t launch_gemm_v8(int dtype, int epi, const GemmArgs& g, hipStream_t s) {
    if (!gemm_v8_ok(dtype, epi, g)) return hipErrorInvalidValue;
    return dtype == DT_BF16 ? launch_v8_t<bf16_t>(epi, g, s) : launch_v8_t<f16_t>(epi, g, s);
}

}  // namespace rz
