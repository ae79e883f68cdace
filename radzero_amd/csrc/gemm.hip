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
#include "gemm_common.h"

namespace rz {


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
    // v7 (gemm7.hip, 16-bit only): 0.755 ms per layer, main loop 1.40 us per K tile against 1.68 for v3 (tools/kslope.py)
    if (variant == 0) {
        const bool big = ok3 && (int64_t)(g.M / BM2) * (g.N / BN3) >= 200;
        variant = !big ? 1 : gemm_v7_ok(Traits<T>::kDType, g) ? 7 : 3;
    }
    if (variant == 7 || variant == 9) {
        if (gemm_v7_ok(Traits<T>::kDType, g)) return launch_gemm_v7(variant, Traits<T>::kDType, epi, g, s);
        variant = ok3 ? 3 : 1;
    }
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
