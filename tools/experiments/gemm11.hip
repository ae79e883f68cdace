// radzero_hip — persistent 256x256x64 GEMM for 16-bit operands, ONE phase per K tile ("v11"): gemm8.hip's structure (8 waves as two
// groups staggered by one barrier, 128x64 per wave, wave-private epilogues, operand stream continuous across output tiles) with the K tile
// no longer cut into four LOAD / MFMA phases of 16 MFMAs but run as ONE LOAD part (all 24 fragment reads of the wave's 128x64x64 block +
// its LDS-DMA pieces of the next K tile) and ONE MFMA part of 64 MFMAs: two barriers per K tile instead of eight.
//
// Why (profiles/r01/gemm_v7_inkernel_stamps.log, profiles/r03/gemm_v10_vs_v8_and_ablations.log): in the four-phase loop the matrix pipe
// of a SIMD is handed back and forth between its two waves once per 16 MFMAs, and every hand-over costs a barrier round trip, an
// lgkmcnt / vmcnt wait and the partner's LOAD issue interfering with the MFMA issue: a phase period of ~790 cycles for 512 cycles of
// MFMA (65 %; PMC: MFMA busy 52 % at 1.9 GHz).  With 64 MFMAs per hand-over the same fixed costs are paid a quarter as often.
// The price: the whole K tile's fragments live in registers at once (24 x 4 = 96 VGPRs beside the 128 accumulators).
//
// Intervals (barrier to barrier), group 1 one barrier behind group 0:
//     interval   2T        2T+1      2T+2      2T+3
//     group 0    LOAD(T)   MFMA(T)   LOAD(T+1) MFMA(T+1)
//     group 1    MFMA(T-1) LOAD(T)   MFMA(T)   LOAD(T+1)
// LOAD(T) of group g: ds_read every fragment of K tile T from stage T & 1, then issue this group's LDS-DMA pieces of K tile T+1 into
// stage (T+1) & 1 — group 0: its own A rows (0-127) and ALL W rows; group 1: its own A rows (128-255) — then lgkmcnt(0), barrier.
// MFMA(T): 64 MFMAs from registers, vmcnt(0) (the pieces issued in LOAD(T): in flight for a whole MFMA part), barrier.
//   RAW  pieces of K tile T+1: group 0's are retired by its vmcnt(0) at the end of interval 2T+1 and a barrier follows; first read in
//        2T+2 (group 0) / 2T+3 (group 1).  Group 1's own A rows: retired at the end of interval 2T+2, read by group 1 only, in 2T+3.
//   WAR  stage (T+1) & 1 held K tile T-1: group 0 read it in interval 2T-2, group 1 in 2T-1, every read retired (lgkmcnt(0)) before the
//        barrier that ends the interval.  Group 0 overwrites W and its own A rows from interval 2T on, group 1 its own A rows (which
//        only it reads) from 2T+1 on.
// Tile seam: LOAD(nk-1) issues the NEXT output tile's K tile 0; the epilogue sits between MFMA(nk-1)'s barrier and LOAD(0), wave-private,
// no barrier inside, so both groups keep their barrier count.  Every vmcnt wait is a plain vmcnt(0): the pieces it must retire are
// YOUNGER than the previous epilogue's stores.  K / 64 >= 2.
// Results are bit-identical to the two-stage kernel (same MFMA order over K per accumulator): tests/test_gpu_kernels.py.
#include <type_traits>

#include "gemm_common.h"
#include "gemm8_epilogue.h"

namespace rz {

constexpr int V11_STAGE = 65536;
constexpr int V11_WAVE_LDS = 4096;

__device__ __forceinline__ void v11_glds(const char* base, unsigned off, char* lds_dst) {
    asm volatile("" : "+s"(base));
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v11(GemmArgs g) {
    static_assert(sizeof(T) == 2, "v11 is for 16-bit operands");
    typedef typename Traits<T>::frag frag_t;
    __shared__ __attribute__((aligned(1024))) char lds[2 * V11_STAGE + 8 * V11_WAVE_LDS];     // 160 KB: one workgroup per CU

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    const int tiles_n = g.N / 256, tiles_m = g.M / 256, ntiles = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, stride = gridDim.x >> 3;
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int cnt = tq + (xcd < tr ? 1 : 0);
    if (slot >= cnt) return;

    const int64_t lda_b = g.lda * 2, ldw_b = g.ldw * 2;
    const int nk = g.K / 64;
    // LDS-DMA: a piece = 8 panel rows (1 KB).  Lane l lands on row +(l >> 3), chunk position l & 7, and fetches chunk
    // (l & 7) ^ swz_std(row) = (l & 7) ^ ((4 p + (l >> 4)) & 7) for piece p: two lane offsets per operand (p even / odd).
    unsigned a_off[2], w_off[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const unsigned sw = (unsigned)(((lane & 7) ^ ((4 * par + (lane >> 4)) & 7)) << 4);
        a_off[par] = (unsigned)((lane >> 3) * lda_b) + sw;
        w_off[par] = (unsigned)((lane >> 3) * ldw_b) + sw;
    }
    // this wave's pieces of every K tile: A rows [128 wr + 32 wc, +32) (4 pieces); group 0 also W rows [64 wc, +64) (8 pieces)
    const int a_row = wr * 128 + wc * 32, w_row = wc * 64;
    const unsigned frd = (unsigned)(l15 * 128 + ((lg ^ ((l15 >> 1) & 7)) << 4));
    const unsigned a_rd = (unsigned)(wr * 128 * 128) + frd;
    const unsigned b_rd = (unsigned)(256 * 128 + wc * 64 * 128) + frd;
    char* wl = lds + 2 * V11_STAGE + wave * V11_WAVE_LDS;

    auto tile_origin = [&](int idx, int& m0, int& n0) {
        int tm, tn;
        tile_coords<4>(lo + idx, tiles_m, tiles_n, tm, tn);
        m0 = tm * 256;
        n0 = tn * 256;
    };
    auto issue = [&](const char* Asrc, const char* Wsrc, char* stage) {      // Asrc / Wsrc: tile origin + K offset
#pragma unroll
        for (int p = 0; p < 4; ++p)
            v11_glds(Asrc + (int64_t)(a_row + 8 * p) * lda_b, a_off[p & 1], stage + (a_row + 8 * p) * 128);
        if (wr == 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                v11_glds(Wsrc + (int64_t)(w_row + 8 * p) * ldw_b, w_off[p & 1], stage + 256 * 128 + (w_row + 8 * p) * 128);
        }
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
    if constexpr (LN_CONSUMER) v8_prefetch_ln(g, wl, m0 + wr * 128, n0 + wc * 64, lane);
    // prologue (once per workgroup): K tile 0 landed in stage 0 for everybody
    issue(Ab, Wb, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    for (;;) {
        const bool has_next = idx + stride < cnt;
        int m1 = m0, n1 = n0;
        if (has_next) tile_origin(idx + stride, m1, n1);
        const char* An = reinterpret_cast<const char*>(g.A) + (int64_t)m1 * lda_b;
        const char* Wn = reinterpret_cast<const char*>(g.W) + (int64_t)n1 * ldw_b;
        const bool vt_tile = (EPI == EPI_VT) || ((EPI == EPI_QKV || EPI == EPI_QKV_LN) && n0 >= g.split_n);

        auto k_loop = [&](auto swap_c) {
            constexpr bool SWAP = decltype(swap_c)::value;
            for (int kt = 0; kt < nk; ++kt) {
                char* cur = lds + (kt & 1) * V11_STAGE;
                char* nxt = lds + ((kt + 1) & 1) * V11_STAGE;
                // ---- LOAD part
                frag_t fa[2][8], fb[2][4];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[ks][j] = *reinterpret_cast<const frag_t*>(cur + ((b_rd ^ (ks * 64)) + j * 2048));
#pragma unroll
                    for (int i = 0; i < 8; ++i) fa[ks][i] = *reinterpret_cast<const frag_t*>(cur + ((a_rd ^ (ks * 64)) + i * 2048));
                }
                // K tile kt+1 of this output tile, or K tile 0 of the next one (behind a workgroup's last tile: a harmless re-fetch)
                const bool in1 = kt + 1 < nk;
                issue(in1 ? Ab + (int64_t)(kt + 1) * 128 : An, in1 ? Wb + (int64_t)(kt + 1) * 128 : Wn, nxt);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragments in registers AND this stage's reads retired (WAR, see header)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA part
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if constexpr (SWAP) acc[a][i][j] = mma(fb[ks][j], fa[ks][a * 4 + i], acc[a][i][j]);
                                else acc[a][i][j] = mma(fa[ks][a * 4 + i], fb[ks][j], acc[a][i][j]);
                            }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the pieces issued in this K tile's LOAD part
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        const int mw = m0 + wr * 128, nw = n0 + wc * 64;
        if (vt_tile) {
            if constexpr (EPI == EPI_VT || EPI == EPI_QKV || EPI == EPI_QKV_LN) {
                k_loop(std::integral_constant<bool, false>{});
                v8_epilogue<T, EPI, false>(g, acc, wl, mw, nw, lane);
            }
        } else {
            if constexpr (EPI != EPI_VT) {
                k_loop(std::integral_constant<bool, true>{});
                v8_epilogue<T, EPI, true>(g, acc, wl, mw, nw, lane);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        if constexpr (LN_CONSUMER) {
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
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();
}

static int v11_grid() {
    static int grid = 0;
    if (grid == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            cus = 256;
        grid = cus / 8 * 8;
    }
    return grid;
}

template <typename T>
static hipError_t launch_v11_t(int epi, const GemmArgs& g, hipStream_t s) {
    dim3 grid(v11_grid()), block(512);
#define RZ_CASE11(E) case E: hipLaunchKernelGGL((gemm_kernel_v11<T, E>), grid, block, 0, s, g); break;
    switch (epi) {
        RZ_CASE11(EPI_STORE)
        RZ_CASE11(EPI_GELU)
        RZ_CASE11(EPI_HEADS)
        RZ_CASE11(EPI_VT)
        RZ_CASE11(EPI_RESID_SCALE)
        RZ_CASE11(EPI_RESID_ADD)
        RZ_CASE11(EPI_PATCH)
        RZ_CASE11(EPI_STORE_F32)
        RZ_CASE11(EPI_QKV)
        RZ_CASE11(EPI_RESID_SCALE_LN)
        RZ_CASE11(EPI_QKV_LN)
        RZ_CASE11(EPI_GELU_LN)
        default: return hipErrorInvalidValue;
    }
#undef RZ_CASE11
    return hipGetLastError();
}

bool gemm_v11_ok(int dtype, int epi, const GemmArgs& g) { return gemm_v8_ok(dtype, epi, g); }

hipError_t launch_gemm_v11(int dtype, int epi, const GemmArgs& g, hipStream_t s) {
    if (!gemm_v11_ok(dtype, epi, g)) return hipErrorInvalidValue;
    return dtype == DT_BF16 ? launch_v11_t<bf16_t>(epi, g, s) : launch_v11_t<f16_t>(epi, g, s);
}

}  // namespace rz
