// radzero_hip — deep-pipelined 256x256x64 GEMM for 16-bit operands ("v7"); same fused epilogues and call sites as
// gemm.hip (C[M,N] = A[M,K] W[N,K]^T, both operands K-contiguous).
//
// EIGHT waves (2x4, 128x64 each, two per SIMD) run as two groups (wr = 0 / 1) staggered by one barrier, four phases per
// K tile.  A phase is
//     LOAD part : ds_read the fragments of one 64x32 accumulator quadrant, issue 2 LDS-DMA pieces (one staging unit)
//     barrier
//     MFMA part : 16 MFMAs (quadrant x K=64); s_waitcnt vmcnt(6)
//     barrier
// and because group 1 runs one barrier behind group 0, the MFMA part of one wave on a SIMD always coincides with the
// LOAD part of the other: the matrix pipe is fed by one wave while the other pays ds_read latency and DMA issue.
// The DMA queue is never drained: three staging units (48 KB per CU) stay in flight across every barrier, and each unit
// has at least 3.5 phases (~2000 cycles) to land — in-kernel stamps (MODE 2, tools/kstamp.py) showed the earlier
// one-unit-deep version of this loop waiting ~40 % of its time for operands from beyond L2.
//
// LDS: two 64 KB K-tile buffers, each an A panel and a W panel of 256 rows x 128 B (swizzle of rz_common.h).
// Staging units of K tile T (16 KB = 16 pieces of 8 rows, 2 per wave) and the phase that issues them:
//     U0(T) = A rows 0-63 of each 128-row half     issued in phase 2 of tile T-2   first read in phase 0 of tile T
//     U1(T) = W rows with (row & 32) == 0          issued in phase 3 of tile T-2   first read in phase 0
//     U2(T) = W rows with (row & 32) != 0          issued in phase 0 of tile T-1   first read in phase 1
//     U3(T) = A rows 64-127 of each half           issued in phase 1 of tile T-1   first read in phase 2
// (tile T-2 lives in the same buffer as T: U0/U1 refill rows of the CURRENT buffer that were read in its phase 0.)
// A half wr is staged and read by group wr only.
// Ordering (phases numbered globally, q = 4T + u; group 1 is one barrier interval late):
//   RAW  a unit issued in LOAD(p) is first read in LOAD(p+5) or later.  Every wave's vmcnt(6) at the end of MFMA(p+3)
//        leaves only the three younger units in flight, i.e. retires its pieces of that unit; group 1 executes that wait
//        in interval 2p+8 and a barrier follows, group 0 reads in interval 2p+10, group 1 in 2p+11.
//   WAR  a unit overwrites rows whose last ds_read was issued at least two phases (four barriers) earlier by either
//        group, and every ds_read is retired (lgkmcnt) before the MFMAs of its own phase.
#include <type_traits>

#include "gemm_common.h"

namespace rz {

constexpr int V7_BM = 256, V7_BN = 256;
constexpr int V7_STAGE = (V7_BM + V7_BN) * 128;     // 64 KB: A panel (256 rows x 128 B) then W panel

template <typename T, bool SWAP>
__device__ __forceinline__ void v7_mma(f32x4& c, const typename Traits<T>::frag& a, const typename Traits<T>::frag& b) {
    if constexpr (SWAP) c = mma(b, a, c); else c = mma(a, b, c);
}

__device__ __forceinline__ void v7_glds(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

#define RZ_STAMP(idx)                                                                                   \
    if constexpr (MODE == 2) {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[idx]) :: "memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    }

template <int N> __device__ __forceinline__ void v7_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

// One K tile.  S1: tile T+1 exists (issue its U2, U3 into `nxt`), S2: tile T+2 exists (issue its U0, U1 into `cur`).
// a1/w1 (a2/w2): this wave's per-lane source pointers at the K offset of tile T+1 (T+2).
// MX: this K tile is a 128-byte fp8 tile of the fp32 mode's MX form (rz_common.h): ONE block-scaled MFMA per accumulator tile, with the
// fragments of both k-halves as its 32-byte operands; sa / sw = this lane's E8M0 scale bytes for the A rows / the W rows.
template <typename T, bool SWAP, bool S1, bool S2, int MODE, bool MX = false>
__device__ __forceinline__ void v7_tile(f32x4 (&acc)[2][4][4], char* cur, char* nxt, unsigned a_rd, unsigned b_rd,
                                        const char* const (&a1)[2], const char* const (&w1)[2], const char* const (&a2)[2],
                                        const char* const (&w2)[2], int64_t a_sub, int64_t w_sub, unsigned a_dst, unsigned w_dst,
                                        unsigned long long (&st)[24], int sa = 0, int sw = 0, bool rt1 = true, bool rt2 = true) {
    // rt1 / rt2 (wave-uniform, run time): AND-ed with S1 / S2 — lets ONE instantiation serve the steady state and the two tail tiles (the MX
    // kernels: a second and third copy of this body made hipcc rename the accumulators between them and spill ~140 VGPRs at the seams)
    typedef typename Traits<T>::frag frag_t;
    frag_t fa[2][4], fb0[2][2], fb1[2][2];          // [k-half][fragment]
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        // ---- LOAD part
        RZ_STAMP(u * 6 + 0)
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
            if (u == 0 && S1 && rt1) v7_glds(w1[e] + w_sub, nxt + w_dst + 32 * 128 + e * 1024);     // U2(T+1)
            if (u == 1 && S1 && rt1) v7_glds(a1[e] + a_sub, nxt + a_dst + 64 * 128 + e * 1024);     // U3(T+1)
            if (u == 2 && S2 && rt2) v7_glds(a2[e], cur + a_dst + e * 1024);                        // U0(T+2)
            if (u == 3 && S2 && rt2) v7_glds(w2[e], cur + w_dst + e * 1024);                        // U1(T+2)
        }
        RZ_STAMP(u * 6 + 1)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        RZ_STAMP(u * 6 + 2)
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
                        v7_mma<T, SWAP>(acc[mi][i][ni * 2 + j], fa[ks][i], ni ? fb1[ks][j] : fb0[ks][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        RZ_STAMP(u * 6 + 3)
        // everything issued three or more phases ago must have landed: count the units issued in phases q-2, q-1, q
        constexpr int kYoung[4] = {(S1 ? 3 : 0), (S1 ? 3 : 0), (S1 ? 2 : 0) + (S2 ? 1 : 0), (S1 ? 1 : 0) + (S2 ? 2 : 0)};
        if (S1 && S2 && !(rt1 && rt2)) {
            // run-time tail of a (true, true) instantiation: the counts of the (true, false) / (false, false) cases
            if (rt1) { if (u < 2) v7_wait_vm<6>(); else if (u == 2) v7_wait_vm<4>(); else v7_wait_vm<2>(); }
            else v7_wait_vm<0>();
        } else {
            if (u == 0) v7_wait_vm<2 * kYoung[0]>();
            if (u == 1) v7_wait_vm<2 * kYoung[1]>();
            if (u == 2) v7_wait_vm<2 * kYoung[2]>();
            if (u == 3) v7_wait_vm<2 * kYoung[3]>();
        }
        RZ_STAMP(u * 6 + 4)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        RZ_STAMP(u * 6 + 5)
    }
}

// OT = type of the outputs (default: the operand type); float / split_f16 = the fp32 mode's hi/lo-split GEMMs (gemm.hip)
// MXK: the fp32 mode's MX form — operand rows are [K f16 | 2 K fp8 bytes], g.K = 2 K counts 128-byte K tiles x 64: the first half of the
// K tiles runs the f16 MFMAs (a_hi b_hi), the second half the block-scaled fp8 MFMA (the two correction terms)
template <typename T, int EPI, int MODE, typename OT = T, bool MXK = false>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v7(GemmArgs g) {
    static_assert(sizeof(T) == 2, "v7 is for 16-bit operands");
    __shared__ __attribute__((aligned(1024))) char lds[2 * V7_STAGE];
    constexpr bool SWAP = (EPI != EPI_VT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    const int tiles_n = g.N / V7_BN, tiles_m = g.M / V7_BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tm, tn;
    tile_coords<4>(bid, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * V7_BM, n0 = tn * V7_BN;
    const int64_t lda_b = g.lda * 2, ldw_b = g.ldw * 2;
    const int nk = g.K / 64;

    // this wave's two DMA pieces (e = 0, 1) of each unit: 8 panel rows each
    //   A units: rows wr*128 + sub*64 + ((wave&3)*2 + e)*8      W units: rows (wave>>1)*64 + sub*32 + ((wave&1)*2 + e)*8
    // lane l lands on row +(l>>3), chunk l&7, and fetches chunk (l&7) ^ swz_std(row) = (l&7) ^ ((4e + (l>>4)) & 7)
    const int a_row = wr * 128 + (wave & 3) * 16, w_row = (wave >> 1) * 64 + (wave & 1) * 16;
    const char* a_src[2];
    const char* w_src[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int64_t sw = (((lane & 7) ^ ((4 * e + (lane >> 4)) & 7)) << 4);
        a_src[e] = reinterpret_cast<const char*>(g.A) + ((int64_t)m0 + a_row + e * 8 + (lane >> 3)) * lda_b + sw;
        w_src[e] = reinterpret_cast<const char*>(g.W) + ((int64_t)n0 + w_row + e * 8 + (lane >> 3)) * ldw_b + sw;
    }
    const int64_t a_sub = 64 * lda_b, w_sub = 32 * ldw_b;
    const unsigned a_dst = (unsigned)(a_row * 128), w_dst = (unsigned)(V7_BM * 128 + w_row * 128);
    const unsigned frd = (unsigned)(l15 * 128 + ((lg ^ ((l15 >> 1) & 7)) << 4));
    const unsigned a_rd = (unsigned)(wr * 128 * 128) + frd;
    const unsigned b_rd = (unsigned)(V7_BM * 128 + wc * 64 * 128) + frd;

    f32x4 acc[2][4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    {
    // prologue: K tile 0 complete in buffer 0; U0, U1 of tile 1 on their way into buffer 1 (nk >= 2)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        v7_glds(a_src[e], lds + a_dst + e * 1024);
        v7_glds(w_src[e], lds + w_dst + e * 1024);
        v7_glds(w_src[e] + w_sub, lds + w_dst + 32 * 128 + e * 1024);
        v7_glds(a_src[e] + a_sub, lds + a_dst + 64 * 128 + e * 1024);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) v7_glds(a_src[e] + 128, lds + V7_STAGE + a_dst + e * 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) v7_glds(w_src[e] + 128, lds + V7_STAGE + w_dst + e * 1024);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    unsigned long long st[24];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 24; ++i) st[i] = 0;
    }
    // E8M0 scale byte of this lane's 32-element block (block index = lane >> 4): A rows = [lo8 | hi8], W rows = [hi8 | lo8]
    const int sa = lg < 2 ? MX_E8_A_LO : MX_E8_A_HI, sw = lg < 2 ? g.mx_w_e8_hi : g.mx_w_e8_lo;      // per weight matrix (api.hip: chosen from its largest |w| when the planes are built)
    const int nkh = MXK ? nk / 2 : nk;          // MXK: nk >= 4 and even
    int kt = 0;
    for (; kt + 2 < nk && kt < nkh; ++kt) {
        char* cur = lds + (kt & 1) * V7_STAGE;
        char* nxt = lds + ((kt + 1) & 1) * V7_STAGE;
        const int64_t k1 = (int64_t)(kt + 1) * 128, k2 = k1 + 128;
        const char* a1[2] = {a_src[0] + k1, a_src[1] + k1};
        const char* w1[2] = {w_src[0] + k1, w_src[1] + k1};
        const char* a2[2] = {a_src[0] + k2, a_src[1] + k2};
        const char* w2[2] = {w_src[0] + k2, w_src[1] + k2};
        v7_tile<T, SWAP, true, true, MODE>(acc, cur, nxt, a_rd, b_rd, a1, w1, a2, w2, a_sub, w_sub, a_dst, w_dst, st);
    }
    if constexpr (MXK) {
        // every MX tile, the last two included, through ONE copy of the body: the tail's "no tile T+1 / T+2" is a run-time flag
        for (; kt < nk; ++kt) {
            char* cur = lds + (kt & 1) * V7_STAGE;
            char* nxt = lds + ((kt + 1) & 1) * V7_STAGE;
            const bool s1 = kt + 1 < nk, s2 = kt + 2 < nk;
            const int64_t k1 = (int64_t)(s1 ? kt + 1 : kt) * 128, k2 = (int64_t)(s2 ? kt + 2 : kt) * 128;
            const char* a1[2] = {a_src[0] + k1, a_src[1] + k1};
            const char* w1[2] = {w_src[0] + k1, w_src[1] + k1};
            const char* a2[2] = {a_src[0] + k2, a_src[1] + k2};
            const char* w2[2] = {w_src[0] + k2, w_src[1] + k2};
            v7_tile<T, SWAP, true, true, MODE == 2 ? 0 : MODE, true>(acc, cur, nxt, a_rd, b_rd, a1, w1, a2, w2, a_sub, w_sub, a_dst, w_dst, st, sa, sw, s1, s2);
        }
    } else {
        char* cur = lds + (kt & 1) * V7_STAGE;
        char* nxt = lds + ((kt + 1) & 1) * V7_STAGE;
        const int64_t k1 = (int64_t)(kt + 1) * 128;
        const char* a1[2] = {a_src[0] + k1, a_src[1] + k1};
        const char* w1[2] = {w_src[0] + k1, w_src[1] + k1};
        unsigned long long st2[24];
        v7_tile<T, SWAP, true, false, MODE == 2 ? 0 : MODE>(acc, cur, nxt, a_rd, b_rd, a1, w1, a1, w1, a_sub, w_sub, a_dst, w_dst, st2);
        v7_tile<T, SWAP, false, false, MODE == 2 ? 0 : MODE>(acc, nxt, cur, a_rd, b_rd, a1, w1, a1, w1, a_sub, w_sub, a_dst, w_dst, st2);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    }
    const int mw = m0 + wr * 128, nw = n0 + wc * 64;
    // 16-bit row-major outputs: whole-tile staging (gemm_common.h); everything else straight from the accumulators
    if constexpr (sizeof(OT) == 2 && (EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_HEADS || EPI == EPI_VT)) {
        __syncthreads();            // every wave is past its last operand read: LDS is free
        gemm_epilogue_tile16<OT, EPI>(g, acc, lds, m0, n0, wr, wc, lane, tid);
    } else if constexpr ((std::is_same<OT, split_f16>::value && (EPI == EPI_GELU || EPI == EPI_HEADS || EPI == EPI_VT)) ||
                         (std::is_same<OT, split_mx>::value && EPI == EPI_GELU) ||
                         (std::is_same<OT, split_mxa>::value && (EPI == EPI_HEADS || EPI == EPI_VT))) {
        __syncthreads();
        gemm_epilogue_tile_split<EPI, std::is_same<OT, split_mx>::value ? 1 : (std::is_same<OT, split_mxa>::value ? 2 : 0)>(g, acc, lds, m0, n0, wr, wc, lane, tid);
    } else {
        gemm_epilogue<OT, EPI>(g, acc[0], mw, nw, l15, lg);
        gemm_epilogue<OT, EPI>(g, acc[1], mw + 64, nw, l15, lg);
    }
}

template <typename T>
static hipError_t launch_v7_t(int variant, int epi, const GemmArgs& g, hipStream_t s) {
    dim3 grid((g.M / V7_BM) * (g.N / V7_BN)), block(512);
#ifdef RZ_EXPERIMENTS      // variant 9: the in-kernel s_memtime stamps of tools/kstamp.py (tools build only)
#define RZ_CASE7(E) case E: if (variant == 9) { if constexpr (E == EPI_STORE) hipLaunchKernelGGL((gemm_kernel_v7<T, E, 2>), grid, block, 0, s, g); } \
                         else hipLaunchKernelGGL((gemm_kernel_v7<T, E, 0>), grid, block, 0, s, g); break;
#else
#define RZ_CASE7(E) case E: hipLaunchKernelGGL((gemm_kernel_v7<T, E, 0>), grid, block, 0, s, g); break;
#endif
    switch (epi) {
        RZ_CASE7(EPI_STORE)
        RZ_CASE7(EPI_GELU)
        RZ_CASE7(EPI_HEADS)
        RZ_CASE7(EPI_VT)
        RZ_CASE7(EPI_RESID_SCALE)
        RZ_CASE7(EPI_RESID_ADD)
        RZ_CASE7(EPI_PATCH)
        RZ_CASE7(EPI_STORE_F32)
        default: return hipErrorInvalidValue;
    }
#undef RZ_CASE7
    return hipGetLastError();
}

// f16 operands (planes side by side along K), fp32 or hi/lo-split outputs: launch_gemm_split_f32out's large shapes
hipError_t launch_gemm_v7_f16_out(int epi, const GemmArgs& g, bool split_out, hipStream_t s) {
    if (!gemm_v7_ok(DT_F16, g)) return hipErrorInvalidValue;
    dim3 grid((g.M / V7_BM) * (g.N / V7_BN)), block(512);
#define RZ_CASE7O(E, OT) case E: hipLaunchKernelGGL((gemm_kernel_v7<f16_t, E, 0, OT>), grid, block, 0, s, g); break;
    if (split_out) {
        switch (epi) {
            RZ_CASE7O(EPI_GELU, split_f16)
            RZ_CASE7O(EPI_HEADS, split_f16)
            RZ_CASE7O(EPI_VT, split_f16)
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (epi) {
            RZ_CASE7O(EPI_STORE, float)
            RZ_CASE7O(EPI_GELU, float)
            RZ_CASE7O(EPI_HEADS, float)
            RZ_CASE7O(EPI_VT, float)
            default: return hipErrorInvalidValue;
        }
    }
#undef RZ_CASE7O
    return hipGetLastError();
}

// fp32 mode, MX form (rz_common.h): operand rows [K f16 | 2 K fp8 bytes], g.lda / g.ldw = 2 K and g.K = 2 K in f16-element units (so the
// staging code sees an ordinary K' = 2 K GEMM).  out_kind 0: fp32 read-modify-write / table epilogues (EPI_RESID_SCALE, EPI_PATCH);
// 1: hi/lo f16 planes for the attention (EPI_HEADS, EPI_VT); 2: the next GEMM's A operand in MX form (EPI_GELU); 3: hi f16 plane + e4m3
// pair plane for the MX attention (EPI_HEADS, EPI_VT).
bool gemm_v7_mx_ok(const GemmArgs& g) {
    return g.M % V7_BM == 0 && g.N % V7_BN == 0 && g.K % 128 == 0 && g.K >= 256;
}
hipError_t launch_gemm_v7_mx(int epi, const GemmArgs& g, int out_kind, hipStream_t s) {
    if (!gemm_v7_mx_ok(g)) return hipErrorInvalidValue;
    dim3 grid((g.M / V7_BM) * (g.N / V7_BN)), block(512);
#define RZ_CASE7M(E, OT) case E: hipLaunchKernelGGL((gemm_kernel_v7<f16_t, E, 0, OT, true>), grid, block, 0, s, g); break;
    if (out_kind == 0) {
        switch (epi) { RZ_CASE7M(EPI_RESID_SCALE, f16_t) RZ_CASE7M(EPI_PATCH, f16_t) default: return hipErrorInvalidValue; }
    } else if (out_kind == 1) {
        switch (epi) { RZ_CASE7M(EPI_HEADS, split_f16) RZ_CASE7M(EPI_VT, split_f16) default: return hipErrorInvalidValue; }
    } else if (out_kind == 3) {
        switch (epi) { RZ_CASE7M(EPI_HEADS, split_mxa) RZ_CASE7M(EPI_VT, split_mxa) default: return hipErrorInvalidValue; }
    } else {
        switch (epi) { RZ_CASE7M(EPI_GELU, split_mx) default: return hipErrorInvalidValue; }
    }
#undef RZ_CASE7M
    return hipGetLastError();
}

// shape contract (checked by the dispatcher in gemm.hip): M % 256 == 0, N % 256 == 0, K % 64 == 0, K >= 128, 16-bit dtype
bool gemm_v7_ok(int dtype, const GemmArgs& g) {
    return dtype != DT_F32 && g.M % V7_BM == 0 && g.N % V7_BN == 0 && g.K % 64 == 0 && g.K >= 128;
}

// variant 7 = production kernel, 9 = the same loop with s_memtime stamps (EPI_STORE only, tools/kstamp.py)
hipError_t launch_gemm_v7(int variant, int dtype, int epi, const GemmArgs& g, hipStream_t s) {
    if (!gemm_v7_ok(dtype, g)) return hipErrorInvalidValue;
    return dtype == DT_BF16 ? launch_v7_t<bf16_t>(variant, epi, g, s) : launch_v7_t<f16_t>(variant, epi, g, s);
}

}  // namespace rz
