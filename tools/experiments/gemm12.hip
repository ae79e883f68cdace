// radzero_hip — "ping-pong" persistent GEMM for 16-bit operands ("v12"): TWO independent 256-thread workgroups per CU, each
// walking its own list of 256 x 128 output tiles with its own 64 KB operand ring, so that one workgroup's epilogue (bias /
// GELU / fp32 residual read-modify-write / fused-LayerNorm copy: 2.4-11.9 us per tile with the matrix pipe idle in gemm8.hip,
// profiles/r02/gemm_v8_inkernel_stamps.log) runs while the OTHER workgroup of the CU holds the matrix pipe.  A SIMD hosts one
// wave of each workgroup; nothing couples the two (no shared barrier, no shared LDS), the SIMD's arbiter interleaves them.
// Reference computation: TF:dinov2/modeling_dinov2.py:199-213 (q|k|v), :246-251 (out-proj), :281-297 (MLP); the epilogues are
// gemm8.hip's (gemm8_epilogue.h: a wave block is 128 rows x 64 columns there and here), so results are bit-identical.
//
// Geometry.  Workgroup tile 256 (M) x 128 (N), K tile 64.  Four waves (wr = wave >> 1, wc = wave & 1), each 128 x 64 =
// acc[2][4][4] (128 accumulator registers), 64 MFMAs (16x16x32) per K tile in four quadrant phases (mi, ni) = (0,0) (0,1) (1,1)
// (1,0) of 16 MFMAs, exactly gemm8's wave program.  LDS per workgroup: 8 ring slots of 8 KB (64 panel rows x 128 B, the
// swizzle of rz_common.h) + 4 x 4 KB wave-private epilogue regions = 80 KB, two workgroups per CU.
//
// Operand stream.  One K tile = 48 KB = six 8 KB chunks, in the order the phases consume them:
//   c0 = A rows of (wr 0, mi 0)   c1 = A (wr 1, mi 0)   c2 = W rows of ni 0 (32 rows of wc 0, then 32 of wc 1)   c3 = W (ni 1)
//   c4 = A (wr 0, mi 1)           c5 = A (wr 1, mi 1)
// Chunk q of the stream (q = 6 t + c) lives in ring slot q mod 8 (`pos` = slot of c0 of the current K tile, advanced by 6 per K
// tile: a run-time scalar, so K / 64 may be any number >= 4).  Each wave moves 2 of a chunk's 8 one-KB pieces (LDS-DMA, swizzle on
// the source address).  Fragments are read ONE PHASE BEFORE the MFMAs that consume them (96 fragment registers: fa0, fa1, fb0, fb1),
// so an LDS round trip hides under the running quadrant.  K tile t:
//   phase 0: read fb0 <- c2, fb1 <- c3          MFMA (0,0) = fa0 x fb0     wait, barrier     issue c4, c5 of t+1 (slots of c2, c3)
//   phase 1: read fa1 <- c4|c5 (own wr)         MFMA (0,1) = fa0 x fb1     wait, barrier     issue c0, c1 of t+2 (slots of c4, c5)
//   phase 2:                                    MFMA (1,1) = fa1 x fb1
//   phase 3: read fa0 <- c0|c1 of K tile t+1    MFMA (1,0) = fa1 x fb0     wait, barrier     issue c2, c3 of t+2 (slots of c0, c1 of t+1)
// Each "wait, barrier" = s_waitcnt vmcnt(8) + lgkmcnt(0) + s_barrier: (i) this wave's pieces of everything the NEXT reads touch have
// landed — the two chunk pairs issued after the previous two barriers (8 instructions) may stay in flight — and after the barrier so
// have every other wave's; (ii) every wave's reads of this phase have returned, so the chunks they touched may be overwritten.  A chunk
// is issued a whole K tile or more before its first read.
// Tile seams.  The last K tile of an output tile does not read ahead (the epilogue needs the registers) and leaves c0 | c1 of the next
// tile's first K tile unread, so it issues nothing after its last barrier; the first K tile reads fa0 itself in phase 0 and issues the
// postponed pair together with its own after that phase's barrier (wait: vmcnt(4) there).  As in gemm8.hip the epilogue's stores count
// in vmcnt too: they are younger than every operand piece issued before the epilogue, so the first K tile's first two waits allow EXTRA
// (half the epilogue's trailing stores) more.  Behind a workgroup's last output tile the stream re-fetches that tile's first chunks
// into slots nobody reads; drained at the end.
#include <type_traits>

#include "gemm_common.h"
#include "gemm8_epilogue.h"

namespace rz {

constexpr int P12_BM = 256, P12_BN = 128;
constexpr int P12_CHUNK = 8192;
constexpr int P12_RING = 8 * P12_CHUNK;
constexpr int P12_WAVE_LDS = 4096;

template <typename T, bool SWAP>
__device__ __forceinline__ void p12_mma(f32x4& c, const typename Traits<T>::frag& a, const typename Traits<T>::frag& b) {
    if constexpr (SWAP) c = mma(b, a, c); else c = mma(a, b, c);
}

__device__ __forceinline__ void p12_glds(const char* base, unsigned off, char* lds_dst) {
    asm volatile("" : "+s"(base));
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void p12_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ char* p12_slot(char* ring, int pos, int c) { return ring + (((pos + c) & 7) * P12_CHUNK); }

// One K tile (header).  w1 / a1: this wave's source bases at the K offset of K tile t+1, a2 / w2: of t+2 (any may belong to the next
// output tile).  fa0 (the A fragments of mi 0) is loaded one phase early, by the previous call.  first: first K tile of an output tile
// (fa0 is read here, and the refill of c0 / c1's slots happens one barrier later); last: last K tile (no read-ahead across the
// epilogue: its 32 registers are the epilogue's); extra: an epilogue's stores are in the queue (first only).
template <typename T, bool SWAP, int EXTRA>
__device__ __forceinline__ void p12_tile(f32x4 (&acc)[2][4][4], typename Traits<T>::frag (&fa0)[2][4], char* ring, int pos, int wr,
                                         unsigned a_rd, unsigned b_rd, const char* a1, const char* w1, const char* a2, const char* w2,
                                         const unsigned (&a_off)[2], const unsigned (&w_off)[2], int64_t a64, int64_t w32, unsigned dst_w,
                                         bool first, bool last, bool extra, bool flip) {
    typedef typename Traits<T>::frag frag_t;
    frag_t fa1[2][4], fb0[2][2], fb1[2][2];
    auto read_a = [&](frag_t (&fa)[2][4], int c) {
        const char* ca = p12_slot(ring, pos, c + wr);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = *reinterpret_cast<const frag_t*>(ca + ((a_rd ^ (ks * 64)) + i * 2048));
    };
    auto read_b = [&](frag_t (&fb)[2][2], int c) {
        const char* cw = p12_slot(ring, pos, c);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[ks][j] = *reinterpret_cast<const frag_t*>(cw + ((b_rd ^ (ks * 64)) + j * 2048));
    };
    auto quadrant = [&](int mi, int ni, const frag_t (&fa)[2][4], const frag_t (&fb)[2][2]) {
        if (flip) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) p12_mma<T, SWAP>(acc[mi][i][ni * 2 + j], fa[ks][i], fb[ks][j]);
        if (flip) __builtin_amdgcn_s_setprio(0);
    };
    auto issue2 = [&](const char* s0, const char* s1, const unsigned (&off)[2], int c) {      // two chunks -> slots pos+c, pos+c+1
#pragma unroll
        for (int e = 0; e < 2; ++e) p12_glds(s0, off[e], p12_slot(ring, pos, c) + dst_w + e * 1024);
#pragma unroll
        for (int e = 0; e < 2; ++e) p12_glds(s1, off[e], p12_slot(ring, pos, c + 1) + dst_w + e * 1024);
    };
    auto sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- phase 0: quadrant (0, 0)
    if (first) read_a(fa0, 0);
    read_b(fb0, 2);
    read_b(fb1, 3);
    quadrant(0, 0, fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    if (first) { if (extra) p12_wait_vm<4 + EXTRA>(); else p12_wait_vm<4>(); }
    else p12_wait_vm<8>();
    sync();
    if (first) issue2(w1, w1 + w32, w_off, 8);                  // c2, c3 of t+1 -> slots of c0, c1 (read just now instead of one phase early)
    issue2(a1 + a64, a1 + 3 * a64, a_off, 10);                  // c4, c5 of t+1 -> slots of c2, c3
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 1: quadrant (0, 1)
    read_a(fa1, 4);
    quadrant(0, 1, fa0, fb1);
    __builtin_amdgcn_sched_barrier(0);
    if (first && extra) p12_wait_vm<8 + EXTRA>(); else p12_wait_vm<8>();
    sync();
    issue2(a2, a2 + 2 * a64, a_off, 12);                        // c0, c1 of t+2 -> slots of c4, c5
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: quadrant (1, 1)
    quadrant(1, 1, fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 3: quadrant (1, 0)
    if (!last) read_a(fa0, 6);                                  // c0 | c1 of t+1
    quadrant(1, 0, fa1, fb0);
    __builtin_amdgcn_sched_barrier(0);
    p12_wait_vm<8>();
    sync();
    if (!last) issue2(w2, w2 + w32, w_off, 14);                 // c2, c3 of t+2 -> slots of c0, c1 of t+1
    __builtin_amdgcn_sched_barrier(0);
}

// Tile order inside an XCD (g.raster):
//   0: gemm8's — the XCD owns a contiguous range of logical ids, (GROUP_M = 4) x tiles_n groups, m fastest.
//   S > 0: "slab walk" — the XCD owns a band of m tiles and walks it once per slab of <= S n tiles (m-major inside the slab), so a
//      slab's weight rows (S x 128 x K x 2 bytes: 1.5 MB at S = 8, K = 768) stay in the XCD's 4 MB L2 across the whole band and only
//      the A panels stream (re-read once per slab).
struct P12Walk {
    int tiles_m, tiles_n, lo, cnt;         // raster 0: id range;   raster > 0: lo = first m tile of the band, cnt = tiles in the band
    int hb, sw;                            // band height, slab width
    int raster;
    __device__ __forceinline__ void origin(int idx, int& m0, int& n0) const {
        int tm, tn;
        if (raster == 0) {
            tile_coords<4>(lo + idx, tiles_m, tiles_n, tm, tn);
        } else {
            const int per_slab = hb * sw;
            const int s = idx / per_slab, r = idx - s * per_slab;
            const int w = min(sw, tiles_n - s * sw);
            const int mm = r / w;
            tm = lo + mm;
            tn = s * sw + (r - mm * w);
        }
        m0 = tm * P12_BM;
        n0 = tn * P12_BN;
    }
};

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel_v12(GemmArgs g) {
    static_assert(sizeof(T) == 2, "v12 is for 16-bit operands");
    __shared__ __attribute__((aligned(1024))) char lds[P12_RING + 4 * P12_WAVE_LDS];     // 80 KB: two workgroups per CU
    constexpr int EXTRA = V8Epi<(EPI == EPI_QKV || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN) ? EPI_HEADS : EPI>::kExtra;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- this workgroup's tile list
    P12Walk walk;
    walk.tiles_n = g.N / P12_BN; walk.tiles_m = g.M / P12_BM; walk.raster = g.raster & 0xff;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, stride = gridDim.x >> 3;
    if (walk.raster == 0) {
        const int ntiles = walk.tiles_m * walk.tiles_n;
        const int tq = ntiles >> 3, tr = ntiles & 7;
        walk.lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
        walk.cnt = tq + (xcd < tr ? 1 : 0);
        walk.hb = walk.sw = 1;
    } else {
        const int hb = (walk.tiles_m + 7) >> 3;
        walk.lo = min(xcd * hb, walk.tiles_m);
        walk.hb = min(hb, walk.tiles_m - walk.lo);
        const int nslab = (walk.tiles_n + walk.raster - 1) / walk.raster;
        walk.sw = (walk.tiles_n + nslab - 1) / nslab;
        walk.cnt = walk.hb * walk.tiles_n;
    }
    const int cnt = walk.cnt;
    if (slot >= cnt) return;
    // EXPERIMENT (g.raster bits 8-9): static issue priority by role, so that the two waves of a SIMD alternate instead of running
    // their MFMA clusters side by side: 1 = role by the wave's slot on its SIMD, 2 = by the workgroup's slot on its CU (HW_REG_HW_ID)
    const int prio_mode = (g.raster >> 8) & 3;
    if (prio_mode) {
        const int role = (prio_mode == 1 ? __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) : __builtin_amdgcn_s_getreg(4 | (16 << 6) | (3 << 11))) & 1;
        if (role) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
    }
    const bool flip = prio_mode == 0;

    const int64_t lda_b = g.lda * 2, ldw_b = g.ldw * 2;
    const int nk = g.K / 64;

    // this wave's two DMA pieces (e = 0, 1) of every chunk: chunk rows 16 * wave + 8 e + (lane >> 3); lane l lands on chunk position
    // l & 7 of its row and fetches chunk (l & 7) ^ swz_std(row) = (l & 7) ^ ((4 e + (l >> 4)) & 7)
    unsigned a_off[2], w_off[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned sw = (unsigned)(((lane & 7) ^ ((4 * e + (lane >> 4)) & 7)) << 4);
        a_off[e] = (unsigned)((e * 8 + (lane >> 3)) * lda_b) + sw;
        w_off[e] = (unsigned)((e * 8 + (lane >> 3)) * ldw_b) + sw;
    }
    // chunk row r of an A chunk (wr', mi') is tile row wr' * 128 + mi' * 64 + r; of a W chunk (ni') tile row (r >> 5) * 64 + ni' * 32 + (r & 31)
    const int64_t a_wave = (int64_t)(16 * wave) * lda_b;
    const int64_t w_wave = (int64_t)((wave >> 1) * 64 + (wave & 1) * 16) * ldw_b;
    const int64_t a64 = 64 * lda_b, w32 = 32 * ldw_b;
    const unsigned dst_w = (unsigned)(2 * wave * 1024);
    const unsigned frd = (unsigned)(l15 * 128 + ((lg ^ ((l15 >> 1) & 7)) << 4));
    const unsigned a_rd = frd;
    const unsigned b_rd = (unsigned)(wc * 4096) + frd;
    char* wl = lds + P12_RING + wave * P12_WAVE_LDS;

    f32x4 acc[2][4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typename Traits<T>::frag fa0[2][4];          // A fragments of (own wr, mi 0): read one phase ahead, alive across K tiles

    int idx = slot, m0, n0;
    walk.origin(idx, m0, n0);
    const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)m0 * lda_b + a_wave;
    const char* Wb = reinterpret_cast<const char*>(g.W) + (int64_t)n0 * ldw_b + w_wave;

    constexpr bool LN_CONSUMER = (EPI == EPI_QKV_LN || EPI == EPI_GELU_LN);
    if constexpr (LN_CONSUMER) v8_prefetch_ln(g, wl, m0 + wr * 128, n0 + wc * 64, lane);      // oldest in the queue
    // prologue (once per workgroup): K tile 0 -> slots 0..5, c0 c1 of K tile 1 -> slots 6, 7, in stream order
    int pos = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) p12_glds(Ab, a_off[e], lds + 0 * P12_CHUNK + dst_w + e * 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) p12_glds(Ab + 2 * a64, a_off[e], lds + 1 * P12_CHUNK + dst_w + e * 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) p12_glds(Wb, w_off[e], lds + 2 * P12_CHUNK + dst_w + e * 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) p12_glds(Wb + w32, w_off[e], lds + 3 * P12_CHUNK + dst_w + e * 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) p12_glds(Ab + a64, a_off[e], lds + 4 * P12_CHUNK + dst_w + e * 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) p12_glds(Ab + 3 * a64, a_off[e], lds + 5 * P12_CHUNK + dst_w + e * 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) p12_glds(Ab + 128, a_off[e], lds + 6 * P12_CHUNK + dst_w + e * 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) p12_glds(Ab + 128 + 2 * a64, a_off[e], lds + 7 * P12_CHUNK + dst_w + e * 1024);
    p12_wait_vm<8>();                                   // c0 .. c3 of K tile 0
    __builtin_amdgcn_s_barrier();

    bool after_epilogue = false;
    for (;;) {
        const bool has_next = idx + stride < cnt;
        int m1 = m0, n1 = n0;
        if (has_next) walk.origin(idx + stride, m1, n1);
        const char* An = reinterpret_cast<const char*>(g.A) + (int64_t)m1 * lda_b + a_wave;
        const char* Wn = reinterpret_cast<const char*>(g.W) + (int64_t)n1 * ldw_b + w_wave;
        const bool vt_tile = (EPI == EPI_VT) || ((EPI == EPI_QKV || EPI == EPI_QKV_LN) && n0 >= g.split_n);

        auto k_loop = [&](auto swap_c) {
            constexpr bool SWAP = decltype(swap_c)::value;
            constexpr bool PEEL = !(EPI == EPI_QKV || EPI == EPI_QKV_LN);      // see gemm8.hip: the merged kernels hold two copies already
            if constexpr (PEEL) {
                p12_tile<T, SWAP, EXTRA>(acc, fa0, lds, pos, wr, a_rd, b_rd, Ab + 128, Wb + 128, Ab + 256, Wb + 256, a_off, w_off, a64, w32, dst_w,
                                         true, false, after_epilogue, flip);
                pos = (pos + 6) & 7;
            }
            for (int kt = PEEL ? 1 : 0; kt < nk; ++kt) {
                const bool in1 = kt + 1 < nk, in2 = kt + 2 < nk;
                const char* a1 = in1 ? Ab + (int64_t)(kt + 1) * 128 : An + (int64_t)(kt + 1 - nk) * 128;
                const char* w1 = in1 ? Wb + (int64_t)(kt + 1) * 128 : Wn + (int64_t)(kt + 1 - nk) * 128;
                const char* a2 = in2 ? Ab + (int64_t)(kt + 2) * 128 : An + (int64_t)(kt + 2 - nk) * 128;
                const char* w2 = in2 ? Wb + (int64_t)(kt + 2) * 128 : Wn + (int64_t)(kt + 2 - nk) * 128;
                p12_tile<T, SWAP, EXTRA>(acc, fa0, lds, pos, wr, a_rd, b_rd, a1, w1, a2, w2, a_off, w_off, a64, w32, dst_w,
                                         PEEL ? false : kt == 0, kt == nk - 1, PEEL ? false : (kt == 0 && after_epilogue), flip);
                pos = (pos + 6) & 7;
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
                v8_epilogue<T, EPI, true>(g, acc, wl, mw, nw, lane);
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the chunks issued past the last output tile
}

static int v12_grid() {
    static int grid = 0;
    if (grid == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            cus = 256;
        grid = cus / 8 * 8 * 2;        // two 80 KB workgroups per CU; a multiple of 8 keeps `b & 7` = XCD label
    }
    return grid;
}

template <typename T>
static hipError_t launch_v12_t(int epi, const GemmArgs& g, hipStream_t s) {
    dim3 grid(v12_grid()), block(256);
#define RZ_CASE12(E) case E: hipLaunchKernelGGL((gemm_kernel_v12<T, E>), grid, block, 0, s, g); break;
    switch (epi) {
        RZ_CASE12(EPI_STORE)
        RZ_CASE12(EPI_GELU)
        RZ_CASE12(EPI_HEADS)
        RZ_CASE12(EPI_VT)
        RZ_CASE12(EPI_RESID_SCALE)
        RZ_CASE12(EPI_RESID_ADD)
        RZ_CASE12(EPI_PATCH)
        RZ_CASE12(EPI_STORE_F32)
        RZ_CASE12(EPI_QKV)
        RZ_CASE12(EPI_RESID_SCALE_LN)
        RZ_CASE12(EPI_QKV_LN)
        RZ_CASE12(EPI_GELU_LN)
        default: return hipErrorInvalidValue;
    }
#undef RZ_CASE12
    return hipGetLastError();
}

// shape contract: M % 256 == 0, N % 128 == 0, K % 64 == 0, K >= 256, 16-bit dtype, per-lane operand offsets < 4 GB;
// EPI_QKV additionally split_n % 128 == 0.
bool gemm_v12_ok(int dtype, int epi, const GemmArgs& g) {
    if (dtype == DT_F32 || g.M % P12_BM || g.N % P12_BN || g.K % 64 || g.K < 256 || g.raster < 0 || (g.raster & 0xff) > 64) return false;
    if ((int64_t)16 * g.lda * 2 >= ((int64_t)1 << 32) || (int64_t)16 * g.ldw * 2 >= ((int64_t)1 << 32)) return false;
    if ((epi == EPI_QKV || epi == EPI_QKV_LN) && (g.split_n % P12_BN || g.split_n <= 0 || g.split_n >= g.N || !g.out2)) return false;
    if ((epi == EPI_QKV_LN || epi == EPI_GELU_LN) && (!g.ln_stat || !g.scale || !g.bias)) return false;
    if (epi == EPI_RESID_SCALE_LN && (g.N != 768 || !g.ln_part || !g.ln_hb || !g.ln_gamma || !g.ln_mu || !g.scale || !g.resid)) return false;
    return true;
}

hipError_t launch_gemm_v12(int dtype, int epi, const GemmArgs& g, hipStream_t s) {
    if (!gemm_v12_ok(dtype, epi, g)) return hipErrorInvalidValue;
    return dtype == DT_BF16 ? launch_v12_t<bf16_t>(epi, g, s) : launch_v12_t<f16_t>(epi, g, s);
}

}  // namespace rz
