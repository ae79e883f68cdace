// radzero_hip — flash attention, 32x32x16-MFMA formulation (attn_variant 2 / 3; the default is attention.hip's
// 16x16x32 kernel, which measures the same within noise in bench.py), gfx950.
//
// Same math and data layout as attention.hip's flash_attn_kernel (softmax_2(Q K^T) V over per-head tensors,
// TF:dinov2/modeling_dinov2.py:153-178), re-tiled to cut vector-ALU issue pressure (rocprofv3 PMC,
// profiles/r01/pmc_attn_and_gemm_v3.txt: VALU busy 63 % vs MFMA busy 43 % of SIMD cycles in the 16x16x32 kernel):
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

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mma32(const bf16x8& a, const bf16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma32(const f16x8& a, const f16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma32(const f32x8& a, const f32x8& b, f32x16 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}

constexpr int FB_QROWS = 128;   // query rows per workgroup (4 waves x 32)
constexpr int FB_KEYS = 64;     // keys per KV tile
constexpr float FB_PSUM_LIMIT = 4096.0f;
#ifndef FB_PRIO
#define FB_PRIO 2
#endif

template <typename T, int NST>
__global__ __launch_bounds__(256, 3) void flash_attn32_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                              const T* __restrict__ vT, T* __restrict__ ctx,
                                                              int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad) {
    typedef typename Traits<T>::frag frag_t;
    constexpr int ES = (int)sizeof(T);
    constexpr int NPAN = 64 * ES / 128;
    constexpr int TILE = NPAN * 64 * 128;
    constexpr int CNT = NPAN * 4;                                   // global_load_lds per wave per stage
    __shared__ __attribute__((aligned(1024))) char lds[2 * NST * TILE];   // K ring | V ring

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r31 = lane & 31, hh = lane >> 5;

    // XCD-aware mapping: all query blocks of one (image, head) pair run on one XCD (its K/V stay in that L2)
    const int nq = n_pad / FB_QROWS;
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

    // Q fragments (B operand): qf[ks] = Q[row q0 + r31][d = 16ks + 8hh .. +7]
    const int q0 = qb * FB_QROWS + wave * 32;
    frag_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const frag_t*>(qbase + (int64_t)(q0 + r31) * 64 + ks * 16 + hh * 8);

    // ---- loop-invariant LDS byte offsets ----
    const int krow = (r31 & 0x13) | ((r31 & 4) << 1) | ((r31 & 8) >> 1);   // bits 2 and 3 swapped
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
        const int key0 = t * FB_KEYS;
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
    f32x16 cinit;                 // every register = -m (accumulator initialiser of the score MFMAs)
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; cinit[i] = 0.f; }
    float mrow = 0.f, lrow = 0.f;

    auto tile = [&](int t, int buf, auto first_c, auto mask_c) {
        constexpr bool FIRST = decltype(first_c)::value, MASK = decltype(mask_c)::value;
        const char* sk = lds + buf * TILE;
        const char* sv = lds + (NST + buf) * TILE;
        // ---- S' = K Q^T - m : 2 key tiles of 32 x 4 k-steps of 16 ----
        frag_t kf[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kf[kt][ks] = lds_read(sk, koff[ks], kt * (32 * 128));
        f32x16 sacc[2];
        // MFMA clusters run at raised priority: a co-resident wave in its exp phase would otherwise hog the
        // vector issue port (arbitration is priority, then age) and starve the matrix pipe
        __builtin_amdgcn_s_setprio(FB_PRIO);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) sacc[kt] = mma32(kf[kt][0], qf[0], cinit);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) sacc[kt] = mma32(kf[kt][ks], qf[ks], sacc[kt]);
        __builtin_amdgcn_s_setprio(0);
        // V^T fragments of the first two 16-key steps: requested now, consumed after the exponentials
        frag_t vfa[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vfa[u][dt] = lds_read(sv, voff[u], dt * (32 * 128));
        asm volatile("" ::: "memory");

        if constexpr (MASK) {
            const int key0 = t * FB_KEYS;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // register r = 8s + 4e + rr  <->  key 32kt + 16s + 8hh + 4e + rr
                    const int key = key0 + 32 * kt + 16 * (r >> 3) + 8 * hh + (r & 7);
                    if (key >= n_valid) sacc[kt][r] = -INFINITY;
                }
        }
        auto row_max = [&]() {
            float m0 = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) m0 = fmaxf(fmaxf(m0, sacc[0][r]), sacc[1][r]);
            return fmaxf(m0, __shfl_xor(m0, 32, 64));
        };
        if constexpr (FIRST) {            // tile 0 always holds >= 1 live key: the max is finite
            const float mx = row_max();
            mrow = mx;
#pragma unroll
            for (int i = 0; i < 16; ++i) { cinit[i] = -mx; sacc[0][i] -= mx; sacc[1][i] -= mx; }
        }
        float p[2][16];
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[kt][r] = __builtin_amdgcn_exp2f(sacc[kt][r]);
                psum += p[kt][r];
            }
        if constexpr (!FIRST) {
            if (__builtin_expect(__any(psum > FB_PSUM_LIMIT), 0)) {       // also true for inf
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
                }
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        p[kt][r] = __builtin_amdgcn_exp2f(sacc[kt][r] - delta);
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
        // ---- O^T += V^T P^T : 4 key steps of 16 x 2 d tiles of 32 ----
        frag_t vfb[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vfb[u][dt] = lds_read(sv, voff[2 + u], dt * (32 * 128));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_setprio(FB_PRIO);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oacc[dt] = mma32(vfa[u][dt], pf[0][u], oacc[dt]);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oacc[dt] = mma32(vfb[u][dt], pf[1][u], oacc[dt]);
        __builtin_amdgcn_s_setprio(0);
    };
    using TrueT = std::integral_constant<bool, true>;
    using FalseT = std::integral_constant<bool, false>;

    const int ntiles = (n_valid + FB_KEYS - 1) / FB_KEYS;
    const bool ragged = (n_valid % FB_KEYS) != 0;
    const int nplain = ragged ? ntiles - 1 : ntiles;     // tiles [0, nplain) need no masking
    // Ring of NST LDS stages with counted vmcnt: tiles up to t+NST-1 are requested before tile t is computed; the
    // end-of-tile wait retires only tile t+1 (the younger ones stay in flight across the raw s_barrier).
    auto wait_keep = [&](int younger) {      // wait until at most `younger` whole stages are outstanding
        younger = younger < 0 ? 0 : (younger > NST - 2 ? NST - 2 : younger);
        if (NST >= 3 && younger == 1) {
            if constexpr (CNT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    static_assert(NST == 2 || NST == 3, "ring depth");
#pragma unroll
    for (int d = 0; d < NST - 1; ++d)
        if (d < ntiles) stage(d, d);
    wait_keep(ntiles - 1);                    // tile 0 landed
    int buf = 0;
    // the three tile flavours are separate call sites (peeled first tile, plain loop, masked last tile): one merged
    // loop body makes hipcc keep all of them live at once and spill (87 VGPRs, 4x slower)
    auto step = [&](int t, auto first_c, auto mask_c) {
        const int ahead = t + NST - 1;
        if (ahead < ntiles) stage(ahead, buf == 0 ? NST - 1 : buf - 1);      // slot of tile t-1: every wave is past it
        tile(t, buf, first_c, mask_c);
        if (t + 1 < ntiles) wait_keep(ntiles - 2 - t);
        buf = buf == NST - 1 ? 0 : buf + 1;
    };
    if (nplain >= 1) step(0, TrueT{}, FalseT{}); else step(0, TrueT{}, TrueT{});
    for (int t = 1; t < nplain; ++t) step(t, FalseT{}, FalseT{});
    if (ragged && ntiles > 1) step(ntiles - 1, FalseT{}, TrueT{});

    // ---- epilogue: O = O^T / l, ctx[(b*n_pad + q)][h*64 + d], d = 32dt + 8i + 4hh + r ----
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

hipError_t launch_flash_attn32(int dtype, const void* q, const void* k, const void* vT, void* ctx,
                               int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad, int ring, hipStream_t s) {
    if (n_pad % FB_QROWS || n_valid <= 0 || n_valid > n_pad || B <= 0 || H <= 0) return hipErrorInvalidValue;
    const int nq = n_pad / FB_QROWS;
    const int pairs = B * H;
    dim3 grid(((pairs + 7) / 8) * 8 * nq), block(256);
#define RZ_FA32(TT, NS) hipLaunchKernelGGL((flash_attn32_kernel<TT, NS>), grid, block, 0, s, (const TT*)q, (const TT*)k, \
                                            (const TT*)vT, (TT*)ctx, qk_batch_stride, B, H, n_valid, n_pad)
    switch (dtype) {
        case DT_F32: RZ_FA32(float, 2); break;
        case DT_BF16: if (ring == 3) RZ_FA32(bf16_t, 3); else RZ_FA32(bf16_t, 2); break;
        case DT_F16: if (ring == 3) RZ_FA32(f16_t, 3); else RZ_FA32(f16_t, 2); break;
        default: return hipErrorInvalidValue;
    }
#undef RZ_FA32
    return hipGetLastError();
}

}  // namespace rz
