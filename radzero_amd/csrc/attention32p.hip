// radzero_hip — flash attention, 32x32x16-MFMA formulation, SOFTWARE-PIPELINED across KV tiles (attn_variant 4), gfx950.
// Same conventions as attention32.hip (read that header first).  Difference: the score MFMAs of tile t+1 are issued
// BEFORE the exponentials of tile t, inside one basic block, so that each wave itself keeps the matrix pipe and the
// vector ALU busy at the same time instead of relying on other waves being in a different phase (measured: co-resident
// waves of this kernel run in near lock-step, SQ_VALU_MFMA_COEXEC_CYCLES is only 35 % of the MFMA-busy cycles).
// Costs a second score accumulator set (32 VGPRs): 2 waves per SIMD, 4-stage K/V ring (K must be one tile ahead of V).
// STATUS (round 1): correct (same tests as the other variants) but NOT faster: hipcc needs 256 VGPRs + 22 spills; measured
// 666 TFLOP/s (tools/kbench.py) vs 820 for the default kernel.  Kept as an opt-in experiment (attn_variant 4).
//
// Same math and data layout as attention.hip's flash_attn_kernel (softmax_2(Q K^T) V over per-head tensors,
// TF:dinov2/modeling_dinov2.py:153-178), re-tiled because that kernel is VALU-ISSUE bound (rocprofv3 PMC,
// profiles/r01/pmc_attn_v2.txt: VALU busy 63 % vs MFMA busy 43 % of SIMD cycles):
//   * v_mfma_f32_32x32x16 does twice the FLOPs per issued instruction of 16x16x32 (each MFMA holds the SIMD's
//     vector issue port for 8 cycles either way);
//   * one query per lane (q = lane&31): one set of row statistics instead of two, one cross-lane exchange;
//   * NO row-max in the common path: P = 2^(s - m) is computed against the current reference m (kept in a
//     persistent accumulator-init register block, so the MFMA chain leaves s - m for free) and the tile's row sum,
//     which is needed anyway, doubles as the overflow detector: only if some lane's partial row sum exceeds 2^12
//     (some P > 2^7..2^12, or inf) does the wave take the rare path that finds the true max from the still-live
//     score registers, re-centres (m, l, O) and recomputes P.  Exact for any input.
//
// Fragment convention ("wide"): lane (rho = lane&31, hh = lane>>5) holds 8 K-contiguous elements at
// k = 16*step + 8*hh of row/col rho.  D layout: lane (col = lane&31) holds rows (reg&3) + 8*(reg>>2) + 4*hh.
// K rows are fed permuted (bits 2 and 3 of the row index swapped) so that lane hh owns the 8 CONTIGUOUS keys
// 16s + 8hh .. +7 of every 16-key step s: registers 8s..8s+7 of the S accumulator, packed pairwise, ARE the P^T
// fragment of step s, and the V^T fragment is one 16-byte LDS read.
#include <type_traits>

#include "rz_common.h"
#include "rz_kernels.h"

namespace rz {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mma32p(const bf16x8& a, const bf16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma32p(const f16x8& a, const f16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma32p(const f32x8& a, const f32x8& b, f32x16 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}


constexpr int FP_QROWS = 128;
constexpr int FP_KEYS = 64;
constexpr float FP_PSUM_LIMIT = 4096.0f;
constexpr int FP_NST = 4;

template <typename T>
__global__ __launch_bounds__(256, 2) void flash_attn32p_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ vT, T* __restrict__ ctx,
                                                               int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad) {
    typedef typename Traits<T>::frag frag_t;
    constexpr int ES = (int)sizeof(T);
    constexpr int NPAN = 64 * ES / 128;
    constexpr int TILE = NPAN * 64 * 128;
    constexpr int NST = FP_NST;
    constexpr int CNT = NPAN * 4;
    __shared__ __attribute__((aligned(1024))) char lds[2 * NST * TILE];   // K ring | V ring

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r31 = lane & 31, hh = lane >> 5;
    const int nq = n_pad / FP_QROWS;
    const int pairs = B * H;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int pair = (jj / nq) * 8 + xcd;
    const int qb = jj % nq;
    if (pair >= pairs) return;
    const int b = pair / H, h = pair % H;

    const T* qbase = q + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64;
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64);
    const char* vbase = reinterpret_cast<const char*>(vT + ((int64_t)pair * 64) * n_pad);
    const int64_t k_ld = 64 * (int64_t)ES;
    const int64_t v_ld = (int64_t)n_pad * ES;

    const int q0 = qb * FP_QROWS + wave * 32;
    frag_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const frag_t*>(qbase + (int64_t)(q0 + r31) * 64 + ks * 16 + hh * 8);

    const int krow = (r31 & 0x13) | ((r31 & 4) << 1) | ((r31 & 8) >> 1);
    const int ksw = (krow >> 1) & 7, vsw = (r31 >> 1) & 7;
    int koff[4][2], voff[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int byte = (16 * u + 8 * hh) * ES + hf * 16;
            const int pan = byte >> 7, chunk = (byte & 127) >> 4;
            koff[u][hf] = pan * (64 * 128) + krow * 128 + ((chunk ^ ksw) << 4);
            voff[u][hf] = pan * (64 * 128) + r31 * 128 + ((chunk ^ vsw) << 4);
        }

    auto stage = [&](int t, int buf) {
        char* sk = lds + buf * TILE;
        char* sv = lds + (NST + buf) * TILE;
        const int key0 = t * FP_KEYS;
#pragma unroll
        for (int p = 0; p < NPAN; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row8 = (wave * 2 + i) * 8;
                glds_rows8(sk + p * (64 * 128) + row8 * 128, kbase + (int64_t)key0 * k_ld + p * 128, k_ld, row8, lane);
                glds_rows8(sv + p * (64 * 128) + row8 * 128, vbase + (int64_t)key0 * ES + p * 128, v_ld, row8, lane);
            }
        }
    };
    auto lds_read = [&](const char* base, const int (&off)[2], int imm) -> frag_t {
        if constexpr (ES == 4) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(base + off[0] + imm);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(base + off[1] + imm);
            return pack8<T>(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
        } else {
            return *reinterpret_cast<const frag_t*>(base + off[0] + imm);
        }
    };

    f32x16 oacc[2];
    f32x16 cinit;
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; cinit[i] = 0.f; }
    float mrow = 0.f, lrow = 0.f;

    // scores of one tile: S' = K Q^T + cinit (= -m)
    auto scores = [&](int buf, f32x16 (&sacc)[2]) {
        const char* sk = lds + buf * TILE;
        frag_t kf[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kf[kt][ks] = lds_read(sk, koff[ks], kt * (32 * 128));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) sacc[kt] = mma32p(kf[kt][0], qf[0], cinit);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) sacc[kt] = mma32p(kf[kt][ks], qf[ks], sacc[kt]);
    };

    // one pipeline step: [scores of tile t+1 -> snext] || [softmax of tile t from scur] ; then O += V(t) P(t)
    auto step = [&](int t, int buf, int buf_next, f32x16 (&scur)[2], f32x16 (&snext)[2], auto first_c, auto mask_c, auto next_c) {
        constexpr bool FIRST = decltype(first_c)::value, MASK = decltype(mask_c)::value, NEXT = decltype(next_c)::value;
        const char* sv = lds + (NST + buf) * TILE;
        if constexpr (NEXT) scores(buf_next, snext);          // issued first: the MFMAs run under the VALU work below
        frag_t vfa[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vfa[u][dt] = lds_read(sv, voff[u], dt * (32 * 128));
        if constexpr (MASK) {
            const int key0 = t * FP_KEYS;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + 32 * kt + 16 * (r >> 3) + 8 * hh + (r & 7);
                    if (key >= n_valid) scur[kt][r] = -INFINITY;
                }
        }
        auto row_max = [&]() {
            float m0 = fmaxf(scur[0][0], scur[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) m0 = fmaxf(fmaxf(m0, scur[0][r]), scur[1][r]);
            return fmaxf(m0, __shfl_xor(m0, 32, 64));
        };
        if constexpr (FIRST) {            // tile 0: establish the reference point (snext was computed against m = 0)
            const float mx = row_max();
            mrow = mx;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                cinit[i] = -mx; scur[0][i] -= mx; scur[1][i] -= mx;
                if constexpr (NEXT) { snext[0][i] -= mx; snext[1][i] -= mx; }
            }
        }
        float p[2][16];
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[kt][r] = __builtin_amdgcn_exp2f(scur[kt][r]);
                psum += p[kt][r];
            }
        if constexpr (!FIRST) {
            if (__builtin_expect(__any(psum > FP_PSUM_LIMIT), 0)) {       // also true for inf
                asm volatile("" ::: "memory");
                const float delta = fmaxf(row_max(), 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                mrow += delta;
                lrow *= alpha;
                psum = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    cinit[i] = -mrow;
                    oacc[0][i] *= alpha;
                    oacc[1][i] *= alpha;
                    if constexpr (NEXT) { snext[0][i] -= delta; snext[1][i] -= delta; }     // already computed against the old m
                }
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        p[kt][r] = __builtin_amdgcn_exp2f(scur[kt][r] - delta);
                        psum += p[kt][r];
                    }
                asm volatile("" ::: "memory");
            }
        }
        lrow += psum;
        frag_t pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                pf[kt][s] = pack8<T>(p[kt][8 * s + 0], p[kt][8 * s + 1], p[kt][8 * s + 2], p[kt][8 * s + 3],
                                     p[kt][8 * s + 4], p[kt][8 * s + 5], p[kt][8 * s + 6], p[kt][8 * s + 7]);
        frag_t vfb[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vfb[u][dt] = lds_read(sv, voff[2 + u], dt * (32 * 128));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oacc[dt] = mma32p(vfa[u][dt], pf[0][u], oacc[dt]);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oacc[dt] = mma32p(vfb[u][dt], pf[1][u], oacc[dt]);
    };
    using TrueT = std::integral_constant<bool, true>;
    using FalseT = std::integral_constant<bool, false>;

    const int ntiles = (n_valid + FP_KEYS - 1) / FP_KEYS;
    const bool ragged = (n_valid % FP_KEYS) != 0;

    // wait until at most `younger` whole stages are still in flight, then barrier
    auto wait_keep = [&](int younger) {
        if (younger >= 2) { if constexpr (CNT == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        else if (younger == 1) { if constexpr (CNT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // Invariant at the top of step t: tiles <= t+1 are in LDS, tile t+2 may be in flight; step t first requests tile t+3
    // into the slot of tile t-1.
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d < ntiles) stage(d, d);
    wait_keep(ntiles >= 3 ? 1 : 0);          // tiles 0 and 1 landed
    f32x16 sA[2], sB[2];
    scores(0, sA);
    int buf = 0;
    auto nxt = [&](int bfr) { return bfr == NST - 1 ? 0 : bfr + 1; };
    auto advance = [&](int t) {               // end of step t: make tile t+2 visible, keep tile t+3 in flight
        if (t + 1 < ntiles) wait_keep(t + 3 < ntiles ? 1 : 0);
        buf = nxt(buf);
    };
    // ---- tile 0 ----
    if (3 < ntiles) stage(3, 3);
    if (ntiles == 1) {
        if (ragged) step(0, 0, 0, sA, sB, TrueT{}, TrueT{}, FalseT{}); else step(0, 0, 0, sA, sB, TrueT{}, FalseT{}, FalseT{});
    } else {
        step(0, 0, 1, sA, sB, TrueT{}, FalseT{}, TrueT{});
    }
    advance(0);
    // ---- tiles 1 .. ntiles-2: scores live alternately in sB / sA ----
    int t = 1;
    for (; t + 1 < ntiles; t += 2) {
        if (t + 3 < ntiles) stage(t + 3, buf == 0 ? NST - 1 : buf - 1);
        step(t, buf, nxt(buf), sB, sA, FalseT{}, FalseT{}, TrueT{});
        advance(t);
        if (t + 2 < ntiles) {                 // second half of the unrolled pair (keeps sA/sB statically named)
            if (t + 4 < ntiles) stage(t + 4, buf == 0 ? NST - 1 : buf - 1);
            step(t + 1, buf, nxt(buf), sA, sB, FalseT{}, FalseT{}, TrueT{});
            advance(t + 1);
        } else {
            // t+1 is the last tile and its scores are in sA
            if (ragged) step(t + 1, buf, 0, sA, sB, FalseT{}, TrueT{}, FalseT{}); else step(t + 1, buf, 0, sA, sB, FalseT{}, FalseT{}, FalseT{});
            t = ntiles;                       // done
            break;
        }
    }
    if (t == ntiles - 1 && ntiles > 1) {      // last tile, scores in sB
        if (ragged) step(t, buf, 0, sB, sA, FalseT{}, TrueT{}, FalseT{}); else step(t, buf, 0, sB, sA, FalseT{}, FalseT{}, FalseT{});
    }

    float l = lrow + __shfl_xor(lrow, 32, 64);
    const float inv = 1.0f / l;
    T* o = ctx + ((int64_t)b * n_pad + q0 + r31) * (H * 64) + h * 64 + hh * 4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<typename Traits<T>::vec4*>(o + dt * 32 + i * 8) =
                pack4<T>(oacc[dt][4 * i] * inv, oacc[dt][4 * i + 1] * inv, oacc[dt][4 * i + 2] * inv, oacc[dt][4 * i + 3] * inv);
}

}  // namespace

hipError_t launch_flash_attn32p(int dtype, const void* q, const void* k, const void* vT, void* ctx,
                                int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad, hipStream_t s) {
    if (n_pad % FP_QROWS || n_valid <= 0 || n_valid > n_pad || B <= 0 || H <= 0) return hipErrorInvalidValue;
    const int nq = n_pad / FP_QROWS;
    const int pairs = B * H;
    dim3 grid(((pairs + 7) / 8) * 8 * nq), block(256);
    switch (dtype) {
        case DT_BF16:
            hipLaunchKernelGGL(flash_attn32p_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)q, (const bf16_t*)k,
                               (const bf16_t*)vT, (bf16_t*)ctx, qk_batch_stride, B, H, n_valid, n_pad);
            break;
        case DT_F16:
            hipLaunchKernelGGL(flash_attn32p_kernel<f16_t>, grid, block, 0, s, (const f16_t*)q, (const f16_t*)k,
                               (const f16_t*)vT, (f16_t*)ctx, qk_batch_stride, B, H, n_valid, n_pad);
            break;
        default: return hipErrorInvalidValue;       // fp32 parity mode uses the unpipelined kernels
    }
    return hipGetLastError();
}

}  // namespace rz
