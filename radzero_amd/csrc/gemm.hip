// radzero_hip — tiled MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogues), gfx950.
//
// Every linear layer on the path is y = x W^T + b with x row-major [M,K] and W row-major [N,K]
// (nn.Linear storage), so BOTH operands are K-contiguous: exactly the MFMA fragment shape.
// Replaces (TF: = transformers/models): TF:dinov2/modeling_dinov2.py:199-213 (q/k/v), :246-251
// (attention.output.dense), :281-297 (fc1/GELU/fc2), :272-278 (LayerScale) and the residual adds of
// :342-380; TF:mpnet/modeling_mpnet.py:131-171,:204-231; the patch-embedding conv
// TF:dinov2/modeling_dinov2.py:139-148 as an im2col GEMM.
//
// Structure (v1, the "128x128 family"): every wave owns a 64x64 output block (4x4 MFMA 16x16 tiles); a workgroup is 2x2 waves (128x128 tile), 2x1 (128x64) or 1x1 (64x64);
// K panels of 128 bytes per step (64 x 16-bit or 32 x f32) in an LDS ring of two or four stages filled by global_load_lds_dwordx4 with the panel XOR swizzle of
// rz_common.h (four stages: three panel pairs in flight behind a counted vmcnt + bare s_barrier; round 6, small grids).  M must be a multiple of 128 (callers pad rows
// per image), N a multiple of 128, K*sizeof(T) a multiple of 128.  gemm_pair_kernel runs two GEMMs over the same rows (a block's q|k and v projections) as one launch.
// fp32 mode: launch_gemm_split_f32out — the f16 kernels over hi/lo-split operands laid side by side along K (three MFMAs per product).
#include <cstring>

#include <algorithm>

#include "gemm_common.h"

namespace rz {


// OT = type of the outputs the epilogue writes (default: the operand type).  <f16_t, EPI, float> = 16-bit operands, fp32 outputs: the
// fp32 mode's hi/lo-split GEMMs (launch_gemm_split_f32out below).
// MXK (round 6; T = f16_t): the fp32 mode's MX form on the 128 x 128 tile (gemm7.hip gemm_kernel_v7 "MXK" is the 256 x 256 form): operand rows are
// [K f16 | K / 64 pair blocks of 128 bytes], g.K = 2 K counts 128-byte panels x 64; the first half of the panels runs the f16 MFMAs (a_hi b_hi), the
// second half ONE block-scaled e4m3 MFMA per accumulator tile (both correction terms), its 32-byte operands = the two 16-byte fragments of the panel.
// Same products, same order as the 256 x 256 kernels: bit-identical to them (tests/test_gpu_model.py forced-variant checks).
// WM x WN = waves along m / n (64 x 64 outputs each): 2 x 2 is the 128 x 128 tile; the smaller geometries spread a GEMM of few tiles over more CUs.
// S = panel pairs in the LDS ring: S - 1 pairs in flight behind a COUNTED vmcnt and a bare s_barrier.  (A __syncthreads() is a fence — the compiler puts
// s_waitcnt vmcnt(0) before its s_barrier, i.e. it waits for every outstanding LDS-DMA — which is what made this round's first ring experiment a no-op
// and its "not latency-bound" conclusion wrong: profiles/NOTEBOOK.md.)  On a small grid nothing hides the global -> LDS round trip (~0.9 us per panel pair with
// one pair in flight) but the workgroup itself.  Same K order and accumulator ownership in every geometry and depth: bit-identical outputs.
template <int N> __device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// One output tile (tm, tn) of the GEMM `g`: the body of gemm_kernel and of gemm_pair_kernel (two GEMMs over the same rows in one launch).
template <typename T, int EPI, typename OT, bool MXK, int WM, int WN, int S>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& g, int tm, int tn, char* lds) {
    constexpr int TBM = 64 * WM, TBN = 64 * WN, PA = TBM * 128, PB = TBN * 128;
    constexpr int NPER = 8 / WN + 8 / WM;      // LDS-DMA instructions per wave and panel pair
    static_assert((S == 2 || S == 4) && (S - 2) * NPER < 64, "ring depth");
    constexpr bool SWAP = (EPI != EPI_VT && EPI != EPI_VT_LN);
    constexpr int KS = 128 / (32 * (int)sizeof(T));  // MFMA k-steps (of 32 elements) per panel
    typedef typename Traits<T>::frag frag_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int l15 = lane & 15, lg = lane >> 4;
    const int m0 = tm * TBM, n0 = tn * TBN;

    const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * sizeof(T);
    const char* Wb = reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * sizeof(T);
    const int64_t lda_b = g.lda * (int64_t)sizeof(T), ldw_b = g.ldw * (int64_t)sizeof(T);
    const int nk = (g.K * (int)sizeof(T)) / 128;
    // MXK: E8M0 scale byte of this lane's 32-element block (block index = lane >> 4): A rows = [lo8 | hi8], W rows = [hi8 | lo8] (per weight matrix, api.hip)
    [[maybe_unused]] const int sa_mx = lg < 2 ? MX_E8_A_LO : MX_E8_A_HI, sw_mx = lg < 2 ? g.mx_w_e8_hi : g.mx_w_e8_lo;
    [[maybe_unused]] const int nkh = MXK ? nk / 2 : nk;

    auto stage = [&](int kt, int buf) {
        char* sa = lds + buf * PA;
        char* sb = lds + S * PA + buf * PB;
        const char* ga = Ab + (int64_t)kt * 128;
        const char* gb = Wb + (int64_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 8 / WN; ++i) {        // A: 8 WM groups of 8 rows over WM WN waves
            const int row8 = (wave * (8 / WN) + i) * 8;
            glds_rows8(sa + row8 * 128, ga, lda_b, row8, lane);
        }
#pragma unroll
        for (int i = 0; i < 8 / WM; ++i) {        // W: 8 WN groups
            const int row8 = (wave * (8 / WM) + i) * 8;
            glds_rows8(sb + row8 * 128, gb, ldw_b, row8, lane);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // every wave waits until all but the `younger` youngest of its panel pairs have landed, then the workgroup meets
    auto ring_wait = [&](int younger) {
        if (S >= 4 && younger >= 2) wait_vm_barrier<2 * NPER>();
        else if (S >= 3 && younger == 1) wait_vm_barrier<NPER>();
        else wait_vm_barrier<0>();
    };
    // prologue: panels 0 .. S-2 requested, panel 0 waited for
#pragma unroll
    for (int p = 0; p < S - 1; ++p)
        if (p < nk) stage(p, p);
    ring_wait(min(nk, S - 1) - 1);

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // panel kt + S - 1 goes into the buffer iteration kt - 1 read (every wave is past that iteration's closing barrier)
        if (kt + S - 1 < nk) stage(kt + S - 1, buf == 0 ? S - 1 : buf - 1);
        // panel kt + 1 must have landed before the next iteration; the pairs requested after it stay in flight
        const int younger = min(nk - 1, kt + S - 1) - (kt + 1);
        const char* sa = lds + buf * PA;
        const char* sb = lds + S * PA + buf * PB;
        buf = buf + 1 == S ? 0 : buf + 1;
        // fragments of k-step ks+1 are requested before the 16 MFMAs of k-step ks (register double buffer),
        // so only the first LDS round trip of a K panel is exposed
        frag_t fa[2][4], fb[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[0][i] = lds_frag<T>(sa, wm * 64 + i * 16 + l15, lg);
            fb[0][i] = lds_frag<T>(sb, wn * 64 + i * 16 + l15, lg);
        }
        if constexpr (MXK) {
            if (kt >= nkh) {          // a pair-block panel: both 16-byte fragments of every row, then 16 block-scaled MFMAs
                static_assert(!MXK || std::is_same<T, f16_t>::value, "the MX form's hi plane is f16");
                if constexpr (std::is_same<T, f16_t>::value) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        fa[1][i] = lds_frag<T>(sa, wm * 64 + i * 16 + l15, 4 + lg);
                        fb[1][i] = lds_frag<T>(sb, wn * 64 + i * 16 + l15, 4 + lg);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (SWAP) acc[i][j] = mma_mx(fb[0][j], fb[1][j], fa[0][i], fa[1][i], acc[i][j], sw_mx, sa_mx);
                            else acc[i][j] = mma_mx(fa[0][i], fa[1][i], fb[0][j], fb[1][j], acc[i][j], sa_mx, sw_mx);
                        }
                }
                ring_wait(younger);
                continue;
            }
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
        ring_wait(younger);
    }

    gemm_epilogue<OT, EPI>(g, acc, m0 + wm * 64, n0 + wn * 64, l15, lg);
}

template <typename T, int EPI, typename OT = T, bool MXK = false, int WM = 2, int WN = 2, int S = 2>
__global__ __launch_bounds__(64 * WM * WN, S == 2 ? 2 : 1) void gemm_kernel(GemmArgs g) {
    constexpr int TBM = 64 * WM, TBN = 64 * WN;
    __shared__ __attribute__((aligned(1024))) char lds[S * (TBM + TBN) * 128];  // A0 .. A(S-1) B0 .. B(S-1)
    if constexpr (sizeof(T) == 4) {      // exact-fp32 instantiations: predicated launch (fp32 mode's overflow guard, rz_kernels.h GemmArgs::run_if)
        if (g.run_if && *g.run_if == 0) return;
    }
    const int tiles_n = g.N / TBN, tiles_m = g.M / TBM;
    int tm, tn;
    tile_coords<8>(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
    gemm_tile_body<T, EPI, OT, MXK, WM, WN, S>(g, tm, tn, lds);
}

// Two GEMMs over the SAME rows (A, M, K) in one launch: the block's q|k projection (ga) and its v projection (gb) where the merged projection of the persistent
// kernel does not apply (small batches, odd row counts).  Rastered as ONE GEMM of ga.N + gb.N columns — the tiles of a row block share their A panels through the
// XCD's L2 whichever GEMM they belong to; a tile's arithmetic is gemm_kernel's: same bits as the two launches.
template <typename T, int EPIA, int EPIB, typename OTA, typename OTB, bool MXK, int WM, int WN, int S>
__global__ __launch_bounds__(64 * WM * WN, S == 2 ? 2 : 1) void gemm_pair_kernel(GemmArgs ga, GemmArgs gb) {
    constexpr int TBM = 64 * WM, TBN = 64 * WN;
    __shared__ __attribute__((aligned(1024))) char lds[S * (TBM + TBN) * 128];
    const int tna = ga.N / TBN, tiles_n = tna + gb.N / TBN, tiles_m = ga.M / TBM;
    int tm, tn;
    tile_coords<8>(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
    if (tn < tna) gemm_tile_body<T, EPIA, OTA, MXK, WM, WN, S>(ga, tm, tn, lds);
    else gemm_tile_body<T, EPIB, OTB, MXK, WM, WN, S>(gb, tm, tn - tna, lds);
}

constexpr int BM2 = 256;

// ---------------------------------------------------------------------------------------------------
// v3: 256x256 tile, 8 waves (2x4), each wave 128x64 (8x4 MFMA tiles, 128 accumulator VGPRs), two LDS stages
// of 64 KB.  Halves the L2->LDS operand traffic per FLOP of the 128x128 kernel (which is what bounds it:
// ~12 TB/s of L2 reads at 86 % hit rate) and lowers LDS reads per MFMA from 0.5 to 0.375.
// Requires M % 256 == 0 and N % 256 == 0.
// ---------------------------------------------------------------------------------------------------
constexpr int BN3 = 256;
constexpr int STAGE3_BYTES = (BM2 + BN3) * 128;   // 64 KB

template <typename T, int EPI, typename OT = T>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v3(GemmArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE3_BYTES];
    if constexpr (sizeof(T) == 4) {      // exact-fp32 instantiations: predicated launch (GemmArgs::run_if)
        if (g.run_if && *g.run_if == 0) return;
    }
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
            glds_rows8(sb + row8 * 128, gb, ldw_b, row8, lane);
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
    const f32x4 (&lo)[4][4] = *reinterpret_cast<const f32x4 (*)[4][4]>(&acc[0]);
    const f32x4 (&hi)[4][4] = *reinterpret_cast<const f32x4 (*)[4][4]>(&acc[4]);
    // LDS-staged 16-byte stores for the 16-bit row-major / per-head / transposed outputs (+9..15 % on those GEMMs),
    // direct epilogue for GELU (VALU-bound) and the fp32 outputs / residual read-modify-write (equal within noise).
    constexpr bool kLds = sizeof(OT) == 2 && (EPI == EPI_STORE || EPI == EPI_HEADS || EPI == EPI_VT);
    if constexpr (!kLds) {
        gemm_epilogue<OT, EPI>(g, lo, m0 + wm * 128, n0 + wn * 64, l15, lg);
        gemm_epilogue<OT, EPI>(g, hi, m0 + wm * 128 + 64, n0 + wn * 64, l15, lg);
        return;
    } else {
        __syncthreads();                    // every wave is done reading operand stages: LDS is free
        char* wlds = lds + wave * (64 * 256);
        gemm_epilogue_lds<OT, EPI>(g, lo, wlds, m0 + wm * 128, n0 + wn * 64, lane);
        gemm_epilogue_lds<OT, EPI>(g, hi, wlds, m0 + wm * 128 + 64, n0 + wn * 64, lane);
    }
}

// Kernel choice: GemmArgs::variant (0 = auto; a forced kernel is used where its shape constraints hold).  No process-wide state
// lives here: the caller (api.hip) resolves the handle's / the process-wide option into every launch.

// 256 x 256 tiles (one persistent workgroup per CU: rounds of 256 tiles) or 128 x 128 ones (two workgroups per CU: rounds of 512 tiles, four times as many
// tiles)?  Round 5: a cost model fitted to an isolated sweep of both kernels over 11 .. 176 row tiles of 256 (tools/ktiles.py,
// profiles/r05/gemm_tile_choice_sweep.txt): time = rounds x (a + b K / 64) us per round with
//   256 x 256: b = 1.21, a = 20 (fp32 residual read-modify-write epilogues: out-proj, fc2, patch), 13 (GELU: fc1), 8 (the others)
//   128 x 128: b = 0.83, a = 8.5 (residual epilogues), 7.5 (the others)
// which reproduces the winner at every measured point (e.g. out-proj: 128 x 128 up to 126 big tiles and again at 264 - 330 and 528, 256 x 256 at 132 - 252 and 396;
// fc2: 256 x 256 from 132 big tiles except 264 - 330).  Rounds 1-4 used "at least 128 big tiles and rounds >= 55 % full", which sent fc2 of 132 - 141 tiles (eight 518^2
// images, one 1536^2 image) to the small kernel (88 against 77 us) and out-proj / fc2 of 315 - 330 tiles to the big one (69 against 56 - 62 us).  From 1024 big tiles on
// (four rounds) the big kernel is taken as before: the model was not fitted there and the fused-LayerNorm epilogues of the large shapes are tuned on it.
static bool big_tiles_pay(const GemmArgs& g, int epi) {
    if (g.M % BM2 || g.M < 4 * BM2 || g.N % BN3) return false;
    const int64_t t256 = (int64_t)(g.M / BM2) * (g.N / BN3);
    if (t256 >= 1024) return true;
    // rounds 1-4's criterion stays a sufficient one: inside the step (tools/step_ab.py, profiles/r05/gemm_tile_choice_in_step.txt) the small kernel's fused-LayerNorm
    // epilogues lose where the isolated sweep's plain ones win (six 1024^2 images, 378 big tiles: 409 against 391 images/s); the model decides the rest
    if (t256 >= 128 && (double)t256 / (double)(((t256 + 255) / 256) * 256) >= 0.55) return true;
    const bool resid = epi == EPI_RESID_SCALE || epi == EPI_RESID_SCALE_LN || epi == EPI_RESID_ADD || epi == EPI_PATCH || epi == EPI_PATCH_LN;
    const bool gelu = epi == EPI_GELU || epi == EPI_GELU_LN;
    const double nk = (double)g.K / 64.0;
    const double t_big = (resid ? 20.0 : gelu ? 13.0 : 8.0) + 1.21 * nk, t_small = (resid ? 8.5 : 7.5) + 0.83 * nk;
    const double c_big = (double)((t256 + 255) / 256) * t_big, c_small = (double)((4 * t256 + 511) / 512) * t_small;
    return c_big <= c_small;
}

// the persistent kernels (the only ones with the merged q|k|v projection and the whole-batch fused-LayerNorm epilogues); 10, 11 and 12
// are retired experiments, compiled with -DRZ_EXPERIMENTS only (they fall back to 8 in the product library)
static bool persistent_variant(int v) { return v == 8 || v == 10 || v == 11 || v == 12; }

// The merged q|k|v projection (EPI_QKV) exists only in the persistent kernel: callers ask first and fall back to the
// separate EPI_HEADS + EPI_VT launches (fp32 mode, small batches, forced variants).
bool gemm_qkv_fused_ok(int dtype, const GemmArgs& g) {
#ifdef RZ_EXPERIMENTS
    if (g.variant == 12) return gemm_v12_ok(dtype, EPI_QKV, g);
#endif
    // auto: the merged launch (N = 3 D) against ITS alternative, the q|k (N = 2 D) and v (N = D) launches of the small kernel (same cost model)
    auto merged_pays = [&]() {
        if (g.M % BM2 || g.M < 4 * BM2 || g.N % BN3) return false;
        const int64_t rt = g.M / BM2, t256 = rt * (g.N / BN3);
        if (t256 >= 1024 || (t256 >= 128 && (double)t256 / (double)(((t256 + 255) / 256) * 256) >= 0.55)) return true;
        const double nk = (double)g.K / 64.0, t_big = 8.0 + 1.21 * nk, t_small = 7.5 + 0.83 * nk;
        const int64_t t_qk = rt * (g.split_n / BN3), t_v = t256 - t_qk;
        const double c_big = (double)((t256 + 255) / 256) * t_big;
        const double c_small = (double)((4 * t_qk + 511) / 512 + (4 * t_v + 511) / 512) * t_small;
        return c_big <= c_small;
    };
    return (g.variant == 0 || persistent_variant(g.variant)) && gemm_v8_ok(dtype, EPI_QKV, g) && (g.variant != 0 || merged_pays());
}

// EPI_PATCH_LN: the persistent kernel or the 128x128 kernel (bit-identical arithmetic, as for the other fused-LayerNorm epilogues)
bool gemm_patch_ln_ok(int dtype, const GemmArgs& g) {
    return dtype != DT_F32 && (g.variant == 0 || g.variant == 1 || persistent_variant(g.variant)) && g.M > 0 && g.M % BM == 0 && g.N == 768 && g.K % 64 == 0 &&
           g.ln_part && g.ln_hb && g.ln_gamma && g.scale && g.out && g.rows_per_image > 0;
}

// May a Dinov2 block run with its LayerNorms fused into the GEMMs (EPI_*_LN)?  Every 16-bit shape does: the persistent kernel
// takes the large ones, the 128x128 kernel the rest, with bit-identical arithmetic (gemm_common.h::gemm_epilogue_ln).
bool gemm_ln_fused_ok(int dtype, int M, int D, int F, int variant) {
    return dtype != DT_F32 && (variant == 0 || variant == 1 || persistent_variant(variant)) && M > 0 && M % BM == 0 && D == 768 && F % BN == 0;
}

// 128 x 128-kernel family: tile geometry and ring depth of a launch (GemmArgs::small_tile = geometry + 10 S forces either; 0 = this rule).  Measured per GEMM inside
// the step from kernel traces (profiles/r06/small_kernel_per_gemm.txt: nine forced combinations x six shapes), n = tiles of 128 x 128:
//   n <= 256  every workgroup can own a CU: the four-stage ring (three panel pairs in flight) on the LARGEST tile that still spreads the work — 128 x 128 above 96
//             tiles (fc2 of two 518^2 images, 132 tiles: 51.7 -> 40.5 us; of one 1024^2 image, 252: 57.3 -> 43.2), 128 x 64 from 40 (one 518^2 image, 72: 51.1 -> 38.1),
//             64 x 64 below (224^2 x 2, 36: 44.2 -> 37.1; the text encoder);
//   n <= 340  two or three workgroups per CU hide the latency instead: two stages, 128 x 64 (fc2 of four 518^2 images, 264: 61.2 -> 54.6 us; the deep rings 77-108);
//   above     two stages, 128 x 128 (smaller tiles only add operand bytes through each XCD's L2: 396 tiles -16 % in the step).
// CUs of the device (the thresholds below were measured on 256: a partitioned or smaller part scales them)
static int cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) ? n : 256;
    }
    return cus;
}
static int small_tile_choice(const GemmArgs& g) {
    const int forced = g.small_tile % 10;
    if (forced) return forced;
    const int64_t n = (int64_t)(g.M / BM) * (g.N / BN) * 256 / cu_count();      // tiles, in units of a 256-CU chip
    if (n <= 256) return n > 96 ? 1 : n >= 40 ? 3 : 2;
    return n <= 340 ? 3 : 1;
}
static int small_stages_choice(const GemmArgs& g, int st) {
    const int forced = g.small_tile / 10;
    if (forced == 2 || forced == 4) return forced;
    const int wgs = st == 2 ? (g.M / 64) * (g.N / 64) : st == 3 ? (g.M / 128) * (g.N / 64) : (g.M / 128) * (g.N / 128);
    const int kb = st == 2 ? 16 : st == 3 ? 24 : 32;      // LDS per stage: the deep ring only where every workgroup is resident at once
    return wgs <= cu_count() * (160 / (kb * 4)) ? 4 : 2;
}

template <typename T, int EPI, typename OT, bool MXK, int S>
static void launch_small_s(int st, const GemmArgs& g, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if (st == 2) { hipLaunchKernelGGL((gemm_kernel<T, EPI, OT, MXK, 1, 1, S>), dim3((g.M / 64) * (g.N / 64)), dim3(64), 0, s, g); return; }
        if (st == 3) { hipLaunchKernelGGL((gemm_kernel<T, EPI, OT, MXK, 2, 1, S>), dim3((g.M / 128) * (g.N / 64)), dim3(128), 0, s, g); return; }
    }
    hipLaunchKernelGGL((gemm_kernel<T, EPI, OT, MXK, 2, 2, S>), dim3((g.M / BM) * (g.N / BN)), dim3(256), 0, s, g);
}
template <typename T, int EPI, typename OT, bool MXK = false>
static void launch_small(int st, const GemmArgs& g, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if (small_stages_choice(g, st) == 4) return launch_small_s<T, EPI, OT, MXK, 4>(st, g, s);
    }
    launch_small_s<T, EPI, OT, MXK, 2>(st, g, s);
}

template <typename T>
static hipError_t launch_gemm_t(int epi, const GemmArgs& g, hipStream_t s) {
    const bool ok3 = (g.M % BM2 == 0) && (g.M >= 4 * BM2) && (g.N % BN3 == 0);
    int variant = g.variant;
    // measured on MI355X (tools/kbench.py, ms per layer of 8 images): v1 0.975, v3 0.825-0.867, v7 0.745-0.755; with fewer
    // than ~200 big tiles (single-image calls) the 128x128 kernel fills the 256 CUs better.
    // v7 (gemm7.hip, 16-bit only): K loop 1.40 us per 256x256x64 step against 1.68 for v3 (tools/kslope.py);
    // v8 (gemm8.hip): the same loop, persistent, operand stream continuous across output tiles.
    if (variant == 0) {
        // the merged q|k|v projection exists in the persistent kernel only: its caller has already weighed it against the two small launches (gemm_qkv_fused_ok)
        const bool big = (epi == EPI_QKV || epi == EPI_QKV_LN) ? true : big_tiles_pay(g, epi);
        variant = !big ? 1 : gemm_v8_ok(Traits<T>::kDType, epi, g) ? 8 : gemm_v7_ok(Traits<T>::kDType, g) ? 7 : 3;
    }
    if ((epi == EPI_QKV || epi == EPI_QKV_LN) && !persistent_variant(variant)) return hipErrorInvalidValue;     // merged projection: persistent kernels only
    if ((epi == EPI_HEADS_LN || epi == EPI_VT_LN) && persistent_variant(variant)) variant = 1;  // its two halves: 128x128 kernel only
    if (epi > EPI_QKV && !persistent_variant(variant)) variant = 1;                              // fused-LayerNorm epilogues: those kernels
    if (epi == EPI_PATCH_LN && persistent_variant(variant)) variant = 8;                          // round 4's epilogue: gemm8 only (the retired experiments never got it)
#ifdef RZ_EXPERIMENTS
    if (variant == 12) {
        if (gemm_v12_ok(Traits<T>::kDType, epi, g)) return launch_gemm_v12(Traits<T>::kDType, epi, g, s);
        variant = 8;
    }
    if (variant == 11) {
        if (gemm_v11_ok(Traits<T>::kDType, epi, g)) return launch_gemm_v11(Traits<T>::kDType, epi, g, s);
        variant = 8;
    }
    if (variant == 10) {
        if (gemm_v10_ok(Traits<T>::kDType, epi, g)) return launch_gemm_v10(Traits<T>::kDType, epi, g, s);
        variant = 8;
    }
#else
    if (variant == 10 || variant == 11 || variant == 12) variant = 8;
#endif
    if (variant == 8) {
        if (gemm_v8_ok(Traits<T>::kDType, epi, g)) return launch_gemm_v8(Traits<T>::kDType, epi, g, s);
        variant = epi > EPI_QKV ? 1 : 7;
    }
#ifndef RZ_EXPERIMENTS
    if (variant == 9) variant = 7;      // the stamped build exists only in the tools library
#endif
    if (variant == 7 || variant == 9) {
        if (gemm_v7_ok(Traits<T>::kDType, g)) return launch_gemm_v7(variant, Traits<T>::kDType, epi, g, s);
        variant = ok3 ? 3 : 1;
    }
    if (variant != 1 && variant != 3) return hipErrorInvalidValue;
    if (variant == 3 && !ok3) variant = 1;
    const int ntiles = (g.M / BM2) * (g.N / BN3);
    dim3 grid(ntiles), block(512);
    const int st = variant == 1 ? small_tile_choice(g) : 0;
#define RZ_CASE(E) \
    case E: if (variant == 3) hipLaunchKernelGGL((gemm_kernel_v3<T, E>), grid, block, 0, s, g); \
            else launch_small<T, E, T>(st, g, s); break;
#define RZ_CASE1(E) case E: if (variant != 1 || sizeof(T) != 2) return hipErrorInvalidValue; \
                       if constexpr (sizeof(T) == 2) launch_small<T, E, T>(st, g, s); break;
    switch (epi) {
        RZ_CASE(EPI_STORE)
        RZ_CASE(EPI_GELU)
        RZ_CASE(EPI_HEADS)
        RZ_CASE(EPI_VT)
        RZ_CASE(EPI_RESID_SCALE)
        RZ_CASE(EPI_RESID_ADD)
        RZ_CASE(EPI_PATCH)
        RZ_CASE(EPI_STORE_F32)
        RZ_CASE1(EPI_RESID_SCALE_LN)
        RZ_CASE1(EPI_GELU_LN)
        RZ_CASE1(EPI_HEADS_LN)
        RZ_CASE1(EPI_VT_LN)
        RZ_CASE1(EPI_PATCH_LN)
        default: return hipErrorInvalidValue;
    }
#undef RZ_CASE
#undef RZ_CASE1
    return hipGetLastError();
}

// ---- two GEMMs over the same rows in one launch (gemm_pair_kernel): the 16-bit modes' q|k + v projections on the 128 x 128 family ----
bool gemm_pair_ok(int dtype, int epi_a, const GemmArgs& ga, int epi_b, const GemmArgs& gb) {
    if (dtype != DT_BF16 && dtype != DT_F16) return false;
    if (!((epi_a == EPI_HEADS_LN && epi_b == EPI_VT_LN) || (epi_a == EPI_HEADS && epi_b == EPI_VT))) return false;
    if (ga.A != gb.A || ga.lda != gb.lda || ga.M != gb.M || ga.K != gb.K || ga.variant != gb.variant || ga.small_tile != gb.small_tile) return false;
    if (ga.M <= 0 || ga.M % BM || ga.N <= 0 || ga.N % BN || gb.N <= 0 || gb.N % BN || (ga.K * 2) % 128 || (ga.lda * 2) % 16 || (ga.ldw * 2) % 16 || (gb.ldw * 2) % 16) return false;
    if (ga.variant == 1) return true;
    return ga.variant == 0 && !big_tiles_pay(ga, epi_a) && !big_tiles_pay(gb, epi_b);      // each would have gone to this family on its own
}
template <typename T, int EPIA, int EPIB, int S, typename OTA = T, typename OTB = T, bool MXK = false>
static void launch_pair_s(int st, const GemmArgs& ga, const GemmArgs& gb, hipStream_t s) {
    const int n = ga.N + gb.N;
    if (st == 2) hipLaunchKernelGGL((gemm_pair_kernel<T, EPIA, EPIB, OTA, OTB, MXK, 1, 1, S>), dim3((ga.M / 64) * (n / 64)), dim3(64), 0, s, ga, gb);
    else if (st == 3) hipLaunchKernelGGL((gemm_pair_kernel<T, EPIA, EPIB, OTA, OTB, MXK, 2, 1, S>), dim3((ga.M / 128) * (n / 64)), dim3(128), 0, s, ga, gb);
    else hipLaunchKernelGGL((gemm_pair_kernel<T, EPIA, EPIB, OTA, OTB, MXK, 2, 2, S>), dim3((ga.M / BM) * (n / BN)), dim3(256), 0, s, ga, gb);
}
template <typename T>
static hipError_t launch_pair_t(int epi_a, const GemmArgs& ga, const GemmArgs& gb, hipStream_t s) {
    GemmArgs whole = ga;                 // geometry and ring depth as for ONE GEMM of ga.N + gb.N columns
    whole.N = ga.N + gb.N;
    const int st = small_tile_choice(whole), S = small_stages_choice(whole, st);
    if (epi_a == EPI_HEADS_LN) { if (S == 4) launch_pair_s<T, EPI_HEADS_LN, EPI_VT_LN, 4>(st, ga, gb, s); else launch_pair_s<T, EPI_HEADS_LN, EPI_VT_LN, 2>(st, ga, gb, s); }
    else { if (S == 4) launch_pair_s<T, EPI_HEADS, EPI_VT, 4>(st, ga, gb, s); else launch_pair_s<T, EPI_HEADS, EPI_VT, 2>(st, ga, gb, s); }
    return hipGetLastError();
}
hipError_t launch_gemm_pair(int dtype, int epi_a, const GemmArgs& ga, int epi_b, const GemmArgs& gb, hipStream_t s) {
    if (!gemm_pair_ok(dtype, epi_a, ga, epi_b, gb)) return hipErrorInvalidValue;
    return dtype == DT_BF16 ? launch_pair_t<bf16_t>(epi_a, ga, gb, s) : launch_pair_t<f16_t>(epi_a, ga, gb, s);
}

// fp32 mode: the same pair for its split forms (both outputs leave as planes: the split attention's operands).  form 0 = three f16 planes along K (ga / gb as for
// launch_gemm_split_f32out: K = 3 K), form 1 = MX (as for launch_gemm_small_mx: K = 2 K); v_kind = output form of the V^T GEMM: 1 hi / lo f16 planes, 3 hi f16 + e4m3 pair plane
bool gemm_pair_f32_ok(int form, const GemmArgs& ga, const GemmArgs& gb, int v_kind) {
    if (ga.A != gb.A || ga.lda != gb.lda || ga.M != gb.M || ga.K != gb.K || ga.variant != gb.variant || ga.small_tile != gb.small_tile) return false;
    if (ga.M <= 0 || ga.M % BM || ga.N <= 0 || ga.N % BN || gb.N <= 0 || gb.N % BN || (ga.K * 2) % 128 || (ga.lda * 2) % 16 || (ga.ldw * 2) % 16 || (gb.ldw * 2) % 16) return false;
    if (ga.variant != 0 && ga.variant != 1) return false;
    if (form == 1) {
        if (!(v_kind == 1 || v_kind == 3) || !gemm_small_mx_ok(EPI_HEADS, 1, ga) || !gemm_small_mx_ok(EPI_VT, v_kind, gb)) return false;
        return ga.variant == 1 || (gemm_small_mx_pays(EPI_HEADS, ga) && gemm_small_mx_pays(EPI_VT, gb));
    }
    return form == 0 && v_kind == 1 && (ga.variant == 1 || (!big_tiles_pay(ga, EPI_HEADS) && !big_tiles_pay(gb, EPI_VT)));
}
hipError_t launch_gemm_pair_f32(int form, const GemmArgs& ga, const GemmArgs& gb, int v_kind, hipStream_t s) {
    if (!gemm_pair_f32_ok(form, ga, gb, v_kind)) return hipErrorInvalidValue;
    GemmArgs whole = ga;
    whole.N = ga.N + gb.N;
    const int st = small_tile_choice(whole), S = small_stages_choice(whole, st);
#define RZ_PAIR(OTB, MXKV) do { if (S == 4) launch_pair_s<f16_t, EPI_HEADS, EPI_VT, 4, split_f16, OTB, MXKV>(st, ga, gb, s); \
                                 else launch_pair_s<f16_t, EPI_HEADS, EPI_VT, 2, split_f16, OTB, MXKV>(st, ga, gb, s); } while (0)
    if (form == 0) RZ_PAIR(split_f16, false);
    else if (v_kind == 3) RZ_PAIR(split_mxa, true);
    else RZ_PAIR(split_f16, true);
#undef RZ_PAIR
    return hipGetLastError();
}

// fp32 mode on the f16 matrix pipe.  The caller has split both operands into f16 planes laid side by side along K:
//   A' = [A_hi | A_lo | A_hi]  (M x 3K),   W' = [W_hi | W_hi | W_lo]  (N x 3K)    =>   A' W'^T = A_hi W_hi + A_lo W_hi + A_hi W_lo,
// i.e. the product of the two 22-bit operands less the 2^-22 lo.lo term, accumulated in fp32 by the ordinary f16 kernels run over
// K' = 3K.  Epilogues that write fp32 anyway (residual read-modify-write, patch table) go to launch_gemm's 16-bit dispatch unchanged;
// the others run here with fp32 outputs (OT = float: exact-erf GELU, fp32 per-head / transposed / row-major stores).
hipError_t launch_gemm_split_f32out(int epi, const GemmArgs& g, hipStream_t s, bool split_out) {
    if (epi == EPI_RESID_SCALE || epi == EPI_RESID_ADD || epi == EPI_PATCH || epi == EPI_STORE_F32) return launch_gemm(DT_F16, epi, g, s);
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.M % BM || g.N % BN || (g.K * 2) % 128 || (g.lda * 2) % 16 || (g.ldw * 2) % 16) return hipErrorInvalidValue;
    const bool v3 = big_tiles_pay(g, epi);
    if (v3 && (g.variant == 0 || g.variant == 7) && gemm_v7_ok(DT_F16, g)) return launch_gemm_v7_f16_out(epi, g, split_out, s);   // the deeper-pipelined K loop
    dim3 grid((g.M / BM2) * (g.N / BN3)), block(512);
    const int st = v3 ? 0 : small_tile_choice(g);
#define RZ_CASE(E, OT) \
    case E: if (v3) hipLaunchKernelGGL((gemm_kernel_v3<f16_t, E, OT>), grid, block, 0, s, g); \
            else launch_small<f16_t, E, OT>(st, g, s); break;
    if (split_out) {          // outputs leave as hi/lo f16 planes (the next split GEMM's A operand, the split attention's q / k / V^T)
        switch (epi) {
            RZ_CASE(EPI_GELU, split_f16)
            RZ_CASE(EPI_HEADS, split_f16)
            RZ_CASE(EPI_VT, split_f16)
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (epi) {
        RZ_CASE(EPI_STORE, float)
        RZ_CASE(EPI_GELU, float)
        RZ_CASE(EPI_HEADS, float)
        RZ_CASE(EPI_VT, float)
        default: return hipErrorInvalidValue;
    }
#undef RZ_CASE
    return hipGetLastError();
}

// fp32 mode, MX form on the 128 x 128 kernel (small shapes: the persistent 256 x 256 kernel leaves most CUs idle — 63 tiles for an N = 768 GEMM of one
// 1024^2 image): out_kind as launch_gemm_v7_mx (0 fp32 RMW / table, 1 hi/lo f16 planes, 2 the next GEMM's MX A operand, 3 hi f16 + e4m3 pair plane: V^T only here)
bool gemm_small_mx_ok(int epi, int out_kind, const GemmArgs& g) {
    if (g.M <= 0 || g.M % BM || g.N % BN || g.K % 128 || g.K < 256) return false;
    if (out_kind == 0) return epi == EPI_RESID_SCALE || epi == EPI_PATCH;
    if (out_kind == 1) return epi == EPI_HEADS || epi == EPI_VT;
    if (out_kind == 2) return epi == EPI_GELU;
    return out_kind == 3 && epi == EPI_VT;
}
// would the 128 x 128 kernel be the faster one for this MX GEMM?  Measured in the step (profiles/r06/f32_small_mx_kernel_step_ab.txt): yes below ~0.4 rounds
// of the persistent kernel's 256 workgroups (one 1024^2 image: v / out-proj / fc2 63 big tiles; one or two 518^2 images; 224^2 up to 8 images:
// +3 ... +14 % in the step), no from there on — the 16-bit kernels' fitted model (big_tiles_pay) would also send 264 tiles (518^2 x 16: -7 %) to it
bool gemm_small_mx_pays(int epi, const GemmArgs& g) {
    (void)epi;
    return (int64_t)((g.M + BM2 - 1) / BM2) * ((g.N + BN3 - 1) / BN3) < 100;      // 126 tiles (1024^2: q|k of one image, N = 768 of two): -0.8 % with the small kernel
}
hipError_t launch_gemm_small_mx(int epi, const GemmArgs& g, int out_kind, hipStream_t s) {
    if (!gemm_small_mx_ok(epi, out_kind, g)) return hipErrorInvalidValue;
    const int st = small_tile_choice(g);
#define RZ_CASEM(E, OT) case E: launch_small<f16_t, E, OT, true>(st, g, s); break;
    if (out_kind == 0) {
        switch (epi) { RZ_CASEM(EPI_RESID_SCALE, f16_t) RZ_CASEM(EPI_PATCH, f16_t) default: return hipErrorInvalidValue; }
    } else if (out_kind == 1) {
        switch (epi) { RZ_CASEM(EPI_HEADS, split_f16) RZ_CASEM(EPI_VT, split_f16) default: return hipErrorInvalidValue; }
    } else if (out_kind == 3) {
        switch (epi) { RZ_CASEM(EPI_VT, split_mxa) default: return hipErrorInvalidValue; }
    } else {
        switch (epi) { RZ_CASEM(EPI_GELU, split_mx) default: return hipErrorInvalidValue; }
    }
#undef RZ_CASEM
    return hipGetLastError();
}

// src fp32 [rows][K] (row stride ld) -> dst f16 [rows][3K]: [hi | lo | hi] (activations) or, w_layout, [hi | hi | lo] (weights)
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ src, int64_t ld, f16_t* __restrict__ dst, int64_t rows,
                                                          int K, int w_layout, unsigned* ovf_flag) {
    const int k4 = K / 4;
    const int64_t total = rows * k4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / k4;
        const int c = (int)(i - r * k4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * ld + c);
        f16x4 h, l;
        split4(v, h, l, ovf_flag);
        f16_t* o = dst + r * 3 * K + c;
        *reinterpret_cast<f16x4*>(o) = h;
        *reinterpret_cast<f16x4*>(o + K) = w_layout ? h : l;
        *reinterpret_cast<f16x4*>(o + 2 * K) = w_layout ? l : h;
    }
}

// src fp32 [rows][K] -> MX form (rz_common.h): row = [hi f16 x K | per 64 columns: activations lo8 x 64, hi8 x 64 | weights hi8 x 64, lo8 x 64]
__global__ __launch_bounds__(256) void split_mx_rows_kernel(const float* __restrict__ src, int64_t ld, char* __restrict__ dst, int64_t rows,
                                                            int K, int weights, unsigned* ovf_flag, float w_hi_scale, float w_lo_scale) {
    const int k4 = K / 4;
    const int64_t total = rows * k4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / k4;
        const int c = (int)(i - r * k4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * ld + c);
        f16x4 h;
        uint32_t lo8, hi8;
        split4_mx(v, h, lo8, hi8, weights ? w_hi_scale : MX_A_HI_SCALE, weights ? w_lo_scale : MX_A_LO_SCALE, ovf_flag);
        char* o = dst + r * 4 * K;
        *reinterpret_cast<f16x4*>(o + 2 * c) = h;
        char* pr = o + mx_pair_off(K, c);
        *reinterpret_cast<uint32_t*>(pr) = weights ? hi8 : lo8;
        *reinterpret_cast<uint32_t*>(pr + 64) = weights ? lo8 : hi8;
    }
}

__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ src, int64_t n, unsigned* __restrict__ out_bits) {
    unsigned best = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) best = max(best, __float_as_uint(src[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, best);
}

hipError_t launch_absmax_bits(const float* src, int64_t n, unsigned* out_bits, hipStream_t s) {
    if (!src || !out_bits || n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, src, n, out_bits);
    return hipGetLastError();
}

hipError_t launch_split3(const float* src, int64_t ld, void* dst, int64_t rows, int K, int w_layout, unsigned* ovf_flag, hipStream_t s, int w_e8_hi) {
    if (rows <= 0 || K <= 0 || K % 4 || ld % 4) return hipErrorInvalidValue;
    const int64_t total = rows * (K / 4);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    if (w_layout >= 2) {
        if (K % 64) return hipErrorInvalidValue;
        if (w_e8_hi < 12 || w_e8_hi > 254) return hipErrorInvalidValue;
        const float w_hi = ldexpf(1.0f, w_e8_hi - 127);
        hipLaunchKernelGGL(split_mx_rows_kernel, dim3(blocks), dim3(256), 0, s, src, ld, (char*)dst, rows, K, w_layout == 3 ? 1 : 0, ovf_flag, w_hi, w_hi * (1.0f / 2048.0f));
    } else {
        hipLaunchKernelGGL(split3_rows_kernel, dim3(blocks), dim3(256), 0, s, src, ld, (f16_t*)dst, rows, K, w_layout, ovf_flag);
    }
    return hipGetLastError();
}

hipError_t launch_gemm(int dtype, int epi, const GemmArgs& g_in, hipStream_t s) {
    const GemmArgs& g = g_in;
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
