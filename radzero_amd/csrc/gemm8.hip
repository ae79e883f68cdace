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
#include "gemm8_epilogue.h"

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
// MX: this K tile is a 128-byte fp8 pair block of the fp32 mode's MX form (rz_common.h; gemm7.hip v7_tile): ONE block-scaled MFMA per accumulator
// tile with the fragments of both k-halves as its 32-byte operands; sa / sw = this lane's E8M0 scale bytes for the A rows / the W rows.
template <typename T, bool SWAP, int EXTRA, bool MX = false>
__device__ __forceinline__ void v8_tile(f32x4 (&acc)[2][4][4], char* cur, char* nxt, unsigned a_rd, unsigned b_rd,
                                        const char* a1, const char* w1, const char* a2, const char* w2,
                                        const unsigned (&a_off)[2], const unsigned (&w_off)[2], int64_t a_sub, int64_t w_sub,
                                        unsigned a_dst, unsigned w_dst, bool extra, int sa = 0, int sw = 0) {
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
        if constexpr (MX) {
            if constexpr (std::is_same<T, f16_t>::value) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4& c = acc[mi][i][ni * 2 + j];
                        const frag_t& b0 = ni ? fb1[0][j] : fb0[0][j];
                        const frag_t& b1 = ni ? fb1[1][j] : fb0[1][j];
                        if constexpr (SWAP) c = mma_mx(b0, b1, fa[0][i], fa[1][i], c, sw, sa);
                        else c = mma_mx(fa[0][i], fa[1][i], b0, b1, c, sa, sw);
                    }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        v8_mma<T, SWAP>(acc[mi][i][ni * 2 + j], fa[ks][i], ni ? fb1[ks][j] : fb0[ks][j]);
        }
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

// STAMP: diagnostic build (tools/kstamp8.py, compiled only with -DRZ_EXPERIMENTS; never launched by the model): every wave sums the 100 MHz real-time ticks it
// spends in K loops and in epilogues and stores them, with the absolute time of its first 24 epilogue starts, into the
// buffer passed as g.out2 — memory no other code of the kernel reads.
// MXK: the fp32 mode's MX form (gemm7.hip gemm_kernel_v7 "MXK"): operand rows are [K f16 | 2 K fp8 bytes], g.K = 2 K counts 128-byte K tiles x 64,
// the first half of a tile's K tiles runs the f16 MFMAs (a_hi b_hi), the second half the block-scaled fp8 MFMA (the two correction terms).
// FORM >= 0 (with MXK): the outputs leave in one of the fp32 mode's split forms (gemm8_epilogue.h v8_epilogue_split).
template <typename T, int EPI, bool STAMP = false, bool MXK = false, int FORM = -1>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v8(GemmArgs g) {
    static_assert(sizeof(T) == 2, "v8 is for 16-bit operands");
    __shared__ __attribute__((aligned(1024))) char lds[2 * V8_STAGE + 8 * V8_WAVE_LDS];     // 160 KB: one workgroup per CU
    constexpr int EXTRA = FORM >= 0 ? 8 : V8Epi<(EPI == EPI_QKV || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN) ? EPI_HEADS : EPI>::kExtra;     // split epilogues: 32 stores per lane, the last 16 nothing but stores

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
#ifdef RZ_EXPERIMENTS
        // tile-walk experiment (option gemm_raster = 100 + G): groups of G row tiles instead of 4 — G = 1: the workgroups of an XCD take
        // the N / 256 column tiles of ONE 256-row A panel back to back (the panel is fetched once per XCD and stays in its L2)
        if (g.raster >= 100) tile_coords_rt(g.raster - 100, lo + idx, tiles_m, tiles_n, tm, tn);
        else
#endif
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
            constexpr bool PEEL = !(EPI == EPI_QKV || EPI == EPI_QKV_LN) && !MXK;      // MXK: two loop bodies already (f16 tiles, fp8 tiles)
            if constexpr (PEEL)
                v8_tile<T, SWAP, EXTRA>(acc, lds, lds + V8_STAGE, a_rd, b_rd, Ab + 128, Wb + 128, Ab + 256, Wb + 256, a_off, w_off, a_sub,
                                        w_sub, a_dst, w_dst, after_epilogue);
            const int nkh = MXK ? nk / 2 : nk;
            for (int kt = PEEL ? 1 : 0; kt < nkh; ++kt) {
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
            if constexpr (MXK) {
                // E8M0 scale byte of this lane's 32-element block (block index = lane >> 4): A rows = [lo8 | hi8], W rows = [hi8 | lo8]
                const int sa = lg < 2 ? MX_E8_A_LO : MX_E8_A_HI, sw = lg < 2 ? g.mx_w_e8_hi : g.mx_w_e8_lo;      // per weight matrix (api.hip: chosen from its largest |w| when the planes are built)
                for (int kt = nkh; kt < nk; ++kt) {
                    char* cur = lds + (kt & 1) * V8_STAGE;
                    char* nxt = lds + ((kt + 1) & 1) * V8_STAGE;
                    const bool in1 = kt + 1 < nk, in2 = kt + 2 < nk;
                    const char* a1 = in1 ? Ab + (int64_t)(kt + 1) * 128 : An + (int64_t)(kt + 1 - nk) * 128;
                    const char* w1 = in1 ? Wb + (int64_t)(kt + 1) * 128 : Wn + (int64_t)(kt + 1 - nk) * 128;
                    const char* a2 = in2 ? Ab + (int64_t)(kt + 2) * 128 : An + (int64_t)(kt + 2 - nk) * 128;
                    const char* w2 = in2 ? Wb + (int64_t)(kt + 2) * 128 : Wn + (int64_t)(kt + 2 - nk) * 128;
                    v8_tile<T, SWAP, EXTRA, true>(acc, cur, nxt, a_rd, b_rd, a1, w1, a2, w2, a_off, w_off, a_sub, w_sub, a_dst, w_dst, false, sa, sw);
                }
            }
        };
        const int mw = m0 + wr * 128, nw = n0 + wc * 64;
        if (vt_tile) {
            if constexpr (EPI == EPI_VT || EPI == EPI_QKV || EPI == EPI_QKV_LN) {
                k_loop(std::integral_constant<bool, false>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (FORM >= 0) v8_epilogue_split<EPI, FORM>(g, acc, wl, mw, nw, lane);
                else v8_epilogue<T, EPI, false>(g, acc, wl, mw, nw, lane);
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
                if constexpr (FORM >= 0) v8_epilogue_split<EPI, FORM>(g, acc, wl, mw, nw, lane);
                else v8_epilogue<T, EPI, true>(g, acc, wl, mw, nw, lane);
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
        RZ_CASE8(EPI_PATCH_LN)
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
    if (epi == EPI_PATCH_LN && (g.N != 768 || !g.ln_part || !g.ln_hb || !g.ln_gamma || !g.scale || !g.out || g.rows_per_image <= 0)) return false;
    return true;
}

// fp32 mode, MX form on the persistent loop (g as for launch_gemm_v7_mx: operand rows of 2 K f16-element units, g.K = 2 K): the epilogues that
// work straight from the accumulators (out_kind 0: EPI_RESID_SCALE, EPI_PATCH)
// out_kind as launch_gemm_v7_mx: 0 = fp32 read-modify-write / table epilogues, 1 = EPI_HEADS -> hi / lo f16 planes, 2 = EPI_GELU -> the MX form,
// 3 = EPI_VT -> hi f16 plane + e4m3 pair plane (the default forms of the fp32 mode; the others stay on gemm7.hip)
bool gemm_v8_mx_ok(int epi, int out_kind, const GemmArgs& g) {
    const bool form = (out_kind == 0 && (epi == EPI_RESID_SCALE || epi == EPI_PATCH)) || (out_kind == 1 && epi == EPI_HEADS) || (out_kind == 2 && epi == EPI_GELU) ||
                      (out_kind == 3 && epi == EPI_VT);
    return form && gemm_v8_ok(DT_F16, epi, g);
}
hipError_t launch_gemm_v8_mx(int epi, int out_kind, const GemmArgs& g, hipStream_t s) {
    if (!gemm_v8_mx_ok(epi, out_kind, g)) return hipErrorInvalidValue;
    dim3 grid(v8_grid()), block(512);
    if (epi == EPI_RESID_SCALE) hipLaunchKernelGGL((gemm_kernel_v8<f16_t, EPI_RESID_SCALE, false, true>), grid, block, 0, s, g);
    else if (epi == EPI_PATCH) hipLaunchKernelGGL((gemm_kernel_v8<f16_t, EPI_PATCH, false, true>), grid, block, 0, s, g);
    else if (epi == EPI_HEADS) hipLaunchKernelGGL((gemm_kernel_v8<f16_t, EPI_HEADS, false, true, 0>), grid, block, 0, s, g);
    else if (epi == EPI_GELU) hipLaunchKernelGGL((gemm_kernel_v8<f16_t, EPI_GELU, false, true, 1>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel_v8<f16_t, EPI_VT, false, true, 2>), grid, block, 0, s, g);
    return hipGetLastError();
}

hipError_t launch_gemm_v8(int dtype, int epi, const GemmArgs& g, hipStream_t s) {
    if (!gemm_v8_ok(dtype, epi, g)) return hipErrorInvalidValue;
    return dtype == DT_BF16 ? launch_v8_t<bf16_t>(epi, g, s) : launch_v8_t<f16_t>(epi, g, s);
}

}  // namespace rz
