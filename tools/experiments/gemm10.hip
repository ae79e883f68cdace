// radzero_hip — persistent 256x256x64 GEMM for 16-bit operands, FOUR waves ("v10"): one wave per SIMD, a 128x128 output block per
// wave, 256 accumulators pinned in a[0:255], the K loop in inline asm with asm-owned registers.
//
// Why: the 8-wave kernels (gemm7 / gemm8.hip: 128x64 per wave, two waves per SIMD) read
// (128 + 64) x 64 B of fragments per 32 MFMAs = 384 B of LDS per MFMA and cross eight barriers per K tile; a 128x128 wave block
// reads (128 + 128) x 64 B per 64 MFMAs = 256 B per MFMA and this loop crosses ONE barrier per K tile.  The chip is power limited in
// these loops, so bytes moved per MFMA are what buys clock.  hipcc cannot be talked into the register allocation
// (round 1: 1 376 v_accvgpr copies + scratch), so the loop is generated text (tools/gen_gemm10_kloop.py -> gemm10_kloop.inc):
//   a[0:255]    accumulators: block (I, J) of the wave's 8 x 8 grid of 16x16 tiles at a[(8 I + J) 4 ..]
//   v[128:255]  two fragment sets (k-step parity) of 8 A + 8 W fragments
//   v[120:127]  LDS read addresses, s[64:84] running source pointers / strides / loop count, m0
// The accumulators leave the asm statement as eight 32-float OUTPUT operands pinned to a[32 I : 32 I + 31] (the operand limit of an asm
// statement is 30: 8 outputs + 21 inputs), so the compiler knows they are live — a first version that listed the a-file as clobbered
// and fetched it with separate v_accvgpr_read statements let hipcc park epilogue spills in AGPRs that had not been fetched yet.
//
// LDS (160 KB, one workgroup per CU): two 64 KB stages (A panel 256 rows x 128 B, then W panel; chunk swizzle of rz_common.h) + 8 KB per
// wave for the epilogues (two 4 KB halves = gemm8.hip's wave-private regions, one per 64-column half of the wave's block).
// LDS-DMA: wave w stages A rows [64 w, 64 w + 64) and W rows [64 w, 64 w + 64) of every K tile: 16 pieces of 8 rows (1 KB).
//
// Ordering inside the loop, per K tile t in stage st = t & 1 (X_t = the one barrier of the tile, between its two k-steps):
//   RAW  pieces of K tile t+2 are issued after X_t and retired by the issuing wave's vmcnt wait in front of X_{t+1}; the first read of
//        that stage (fragments of k-step 0 of tile t+2) is issued after X_{t+1}'s barrier by every wave.
//   WAR  the pieces of tile t+2 overwrite stage st, whose last reads (fragments of tile t's k-step 1) every wave retired with the
//        lgkmcnt(0) in front of X_t.
//   Tile seam: the last two K tiles of an output tile issue the NEXT output tile's K tiles 0 and 1 (the stream never stops); K tile 1's
//        pieces are still in flight across the epilogue and are OLDER than its stores, so the next tile's X_0 waits with
//        vmcnt(EXTRA) — EXTRA <= the number of 16-byte stores the epilogue issues last; fewer than are really younger is always safe.
//   K / 64 must be even and >= 4.
// Results are bit-identical to the two-stage kernel (same MFMA order over K per accumulator): tests/test_gpu_kernels.py.
#include <type_traits>

#include "gemm_common.h"
#include "gemm8_epilogue.h"
#include "gemm10_kloop.inc"

namespace rz {

constexpr int V10_STAGE = 65536;
constexpr int V10_WAVE_LDS = 8192;

template <typename T, bool SWAP> struct V10Text;
// clang-format off
#define RZ_V10_OPERANDS                                                                                                             \
    : "={a[0:31]}"(r0), "={a[32:63]}"(r1), "={a[64:95]}"(r2), "={a[96:127]}"(r3), "={a[128:159]}"(r4), "={a[160:191]}"(r5),           \
      "={a[192:223]}"(r6), "={a[224:255]}"(r7)                                                                                       \
    : [va] "v"(va), [vw] "v"(vw), [oa0] "v"(oa0), [oa1] "v"(oa1), [ow0] "v"(ow0), [ow1] "v"(ow1),                                    \
      [ablo] "s"(ablo), [abhi] "s"(abhi), [wblo] "s"(wblo), [wbhi] "s"(wbhi), [anlo] "s"(anlo), [anhi] "s"(anhi),                      \
      [wnlo] "s"(wnlo), [wnhi] "s"(wnhi), [sa8] "s"(sa8), [sw8] "s"(sw8), [ldsa] "s"(ldsa), [ldsw] "s"(ldsw), [nk] "s"(nk),           \
      [after] "s"(after), [extra] "n"(EXTRA)
#define RZ_V10_CLOBBERS                                                                                                             \
    : "memory", "scc", "m0", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78",  \
      "s79", "s80", "s81", "s82", "s83", "s84",                                                                                     \
      "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", \
      "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", \
      "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", \
      "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", \
      "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", \
      "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", \
      "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", \
      "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", \
      "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
// clang-format on

// The whole K loop of one output tile (see header).  All scalar operands are wave-uniform.
// Row I of the wave's 8 x 8 grid of 16x16 accumulator tiles = one asm OUTPUT of 32 floats pinned to a[32 I : 32 I + 31] (tile (I, J) =
// elements 4 J .. 4 J + 3): the compiler knows the values are live and where they are, fetches them with v_accvgpr_read as the
// epilogue needs them, and cannot spill over them.
typedef float f32x32 __attribute__((ext_vector_type(32)));

template <typename T, bool SWAP, int EXTRA>
__device__ __forceinline__ void v10_kloop(f32x32& r0, f32x32& r1, f32x32& r2, f32x32& r3, f32x32& r4, f32x32& r5, f32x32& r6, f32x32& r7, unsigned va, unsigned vw, unsigned oa0, unsigned oa1, unsigned ow0, unsigned ow1, unsigned ablo,
                                          unsigned abhi, unsigned wblo, unsigned wbhi, unsigned anlo, unsigned anhi, unsigned wnlo,
                                          unsigned wnhi, unsigned sa8, unsigned sw8, unsigned ldsa, unsigned ldsw, unsigned nk, unsigned after) {
    if constexpr (std::is_same<T, bf16_t>::value) {
        if constexpr (SWAP) asm volatile(RZ_V10_KLOOP_BF16_SWAP RZ_V10_OPERANDS RZ_V10_CLOBBERS);
        else asm volatile(RZ_V10_KLOOP_BF16_PLAIN RZ_V10_OPERANDS RZ_V10_CLOBBERS);
    } else {
        if constexpr (SWAP) asm volatile(RZ_V10_KLOOP_F16_SWAP RZ_V10_OPERANDS RZ_V10_CLOBBERS);
        else asm volatile(RZ_V10_KLOOP_F16_PLAIN RZ_V10_OPERANDS RZ_V10_CLOBBERS);
    }
}

// the 128x64 half `HALF` of the wave's block as gemm8.hip's accumulator array: acc[a][i][j] = tile (I = 4 a + i, J = 4 HALF + j)
template <int HALF>
__device__ __forceinline__ void v10_half(f32x4 (&acc)[2][4][4], const f32x32 (&r)[8]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = (HALF * 4 + j) * 4;
                acc[a][i][j] = (f32x4){r[a * 4 + i][e], r[a * 4 + i][e + 1], r[a * 4 + i][e + 2], r[a * 4 + i][e + 3]};
            }
}

__device__ __forceinline__ void v10_glds(const char* base, unsigned off, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                     (__attribute__((address_space(3))) void*)(uintptr_t)lds_addr, 16, 0, 0);
}

template <typename T, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_kernel_v10(GemmArgs g) {
    static_assert(sizeof(T) == 2, "v10 is for 16-bit operands");
    __shared__ __attribute__((aligned(1024))) char lds[2 * V10_STAGE + 4 * V10_WAVE_LDS];     // 160 KB: one workgroup per CU
    constexpr int EXTRA = V8Epi<(EPI == EPI_QKV || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN) ? EPI_HEADS : EPI>::kExtra;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- this workgroup's tile list (as gemm8.hip: XCD x owns the logical ids of xcd_remap's range x)
    const int tiles_n = g.N / 256, tiles_m = g.M / 256, ntiles = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, stride = gridDim.x >> 3;
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int cnt = tq + (xcd < tr ? 1 : 0);
    if (slot >= cnt) return;

    const int64_t lda_b = g.lda * 2, ldw_b = g.ldw * 2;
    const unsigned nk = (unsigned)(g.K / 64);
    // LDS-DMA: lane l of piece q lands on row 8 q + (l >> 3), chunk position l & 7, and fetches chunk (l & 7) ^ swz_std(row)
    //          = (l & 7) ^ ((4 q + (l >> 4)) & 7): two lane offsets per operand (q even / odd); 8 q rows go into the scalar base
    unsigned oa[2], ow[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const unsigned sw = (unsigned)(((lane & 7) ^ ((4 * par + (lane >> 4)) & 7)) << 4);
        oa[par] = (unsigned)((lane >> 3) * lda_b) + sw;
        ow[par] = (unsigned)((lane >> 3) * ldw_b) + sw;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned ldsa = lds0 + (unsigned)(wave * 64 * 128), ldsw = lds0 + 32768u + (unsigned)(wave * 64 * 128);
    const unsigned frd = (unsigned)(l15 * 128 + ((lg ^ ((l15 >> 1) & 7)) << 4));
    const unsigned va = lds0 + (unsigned)(wr * 128 * 128) + frd, vw = lds0 + 32768u + (unsigned)(wc * 128 * 128) + frd;
    char* wl = lds + 2 * V10_STAGE + wave * V10_WAVE_LDS;

    auto tile_origin = [&](int idx, int& m0, int& n0) {
        int tm, tn;
        tile_coords<4>(lo + idx, tiles_m, tiles_n, tm, tn);
        m0 = tm * 256;
        n0 = tn * 256;
    };
    int idx = slot, m0, n0;
    tile_origin(idx, m0, n0);
    // this wave's staging rows start 64 w rows into the tile's panels
    const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)(m0 + wave * 64) * lda_b;
    const char* Wb = reinterpret_cast<const char*>(g.W) + (int64_t)(n0 + wave * 64) * ldw_b;

    constexpr bool LN_CONSUMER = (EPI == EPI_QKV_LN || EPI == EPI_GELU_LN);
    if constexpr (LN_CONSUMER) {
        v8_prefetch_ln(g, wl, m0 + wr * 128, n0 + wc * 128, lane);
        v8_prefetch_ln(g, wl + 4096, m0 + wr * 128, n0 + wc * 128 + 64, lane);
    }
    // prologue (once per workgroup): K tiles 0 and 1 on their way, K tile 0 landed
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v10_glds(Ab + (int64_t)q * 8 * lda_b + t * 128, oa[q & 1], ldsa + t * V10_STAGE + q * 1024);
            v10_glds(Wb + (int64_t)q * 8 * ldw_b + t * 128, ow[q & 1], ldsw + t * V10_STAGE + q * 1024);
        }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    unsigned after = 0;
    for (;;) {
        const bool has_next = idx + stride < cnt;
        int m1 = m0, n1 = n0;
        if (has_next) tile_origin(idx + stride, m1, n1);
        // behind a workgroup's last output tile the loop re-fetches that tile's first two K tiles into buffers nobody reads (drained below)
        const char* An = reinterpret_cast<const char*>(g.A) + (int64_t)(m1 + wave * 64) * lda_b;
        const char* Wn = reinterpret_cast<const char*>(g.W) + (int64_t)(n1 + wave * 64) * ldw_b;
        const bool vt_tile = (EPI == EPI_VT) || ((EPI == EPI_QKV || EPI == EPI_QKV_LN) && n0 >= g.split_n);
        const uint64_t ab = (uint64_t)Ab, wb = (uint64_t)Wb, an = (uint64_t)An, wn = (uint64_t)Wn;
        const int mw = m0 + wr * 128, nw = n0 + wc * 128;
        auto run = [&](auto swap_c) {
            constexpr bool SWAP = decltype(swap_c)::value;
            f32x32 r[8];
            v10_kloop<T, SWAP, EXTRA>(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], va, vw, oa[0], oa[1], ow[0], ow[1], (unsigned)ab, (unsigned)(ab >> 32), (unsigned)wb, (unsigned)(wb >> 32),
                                      (unsigned)an, (unsigned)(an >> 32), (unsigned)wn, (unsigned)(wn >> 32), (unsigned)(8 * lda_b),
                                      (unsigned)(8 * ldw_b), ldsa, ldsw, nk, after);
            f32x4 acc[2][4][4];
            v10_half<0>(acc, r);
            v8_epilogue<T, EPI, SWAP>(g, acc, wl, mw, nw, lane);
            v10_half<1>(acc, r);
            v8_epilogue<T, EPI, SWAP>(g, acc, wl + 4096, mw, nw + 64, lane);
        };
        if (vt_tile) {
            if constexpr (EPI == EPI_VT || EPI == EPI_QKV || EPI == EPI_QKV_LN) run(std::integral_constant<bool, false>{});
        } else {
            if constexpr (EPI != EPI_VT) run(std::integral_constant<bool, true>{});
        }
        if (!has_next) break;
        if constexpr (LN_CONSUMER) {       // the epilogues above have read their vectors: fetch the next tile's (youngest in the queue)
            v8_prefetch_ln(g, wl, m1 + wr * 128, n1 + wc * 128, lane);
            v8_prefetch_ln(g, wl + 4096, m1 + wr * 128, n1 + wc * 128 + 64, lane);
        }
        idx += stride;
        m0 = m1; n0 = n1;
        Ab = An; Wb = Wn;
        after = 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the pieces issued past the last output tile
}

static int v10_grid() {
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
static hipError_t launch_v10_t(int epi, const GemmArgs& g, hipStream_t s) {
    dim3 grid(v10_grid()), block(256);
#define RZ_CASE10(E) case E: hipLaunchKernelGGL((gemm_kernel_v10<T, E>), grid, block, 0, s, g); break;
    switch (epi) {
        RZ_CASE10(EPI_STORE)
        RZ_CASE10(EPI_GELU)
        RZ_CASE10(EPI_HEADS)
        RZ_CASE10(EPI_VT)
        RZ_CASE10(EPI_RESID_SCALE)
        RZ_CASE10(EPI_RESID_ADD)
        RZ_CASE10(EPI_PATCH)
        RZ_CASE10(EPI_STORE_F32)
        RZ_CASE10(EPI_QKV)
        RZ_CASE10(EPI_RESID_SCALE_LN)
        RZ_CASE10(EPI_QKV_LN)
        RZ_CASE10(EPI_GELU_LN)
        default: return hipErrorInvalidValue;
    }
#undef RZ_CASE10
    return hipGetLastError();
}

// shape contract = gemm8.hip's (M, N multiples of 256; K a multiple of 128, >= 256; per-lane operand offsets < 4 GB; 8 rows of either
// operand < 4 GB apart)
bool gemm_v10_ok(int dtype, int epi, const GemmArgs& g) { return gemm_v8_ok(dtype, epi, g); }

hipError_t launch_gemm_v10(int dtype, int epi, const GemmArgs& g, hipStream_t s) {
    if (!gemm_v10_ok(dtype, epi, g)) return hipErrorInvalidValue;
    return dtype == DT_BF16 ? launch_v10_t<bf16_t>(epi, g, s) : launch_v10_t<f16_t>(epi, g, s);
}

}  // namespace rz
