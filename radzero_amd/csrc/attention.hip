// radzero_hip — attention kernels, gfx950.
//
// flash_attn_kernel: softmax(Q K^T) V for the 14 Dinov2 blocks without ever materialising the
//   (B,12,N,N) score tensor (1.36 GB fp32 per image at 1024^2).  Replaces
//   TF:dinov2/modeling_dinov2.py:153-178 (eager_attention_forward) as called from :182-234.
// flash_attn_split_kernel: the same contraction for the fp32 mode — operands as hi / lo f16 planes, three f16 MFMAs per product.
// text_attn_kernel: MPNet self-attention with additive relative-position bias and key-padding mask,
//   TF:mpnet/modeling_mpnet.py:131-171 (bias computed once per forward, :312-348).
//
// Flash kernel data flow (per workgroup = 128 query rows of one (image, head); 4 waves x 32 rows):
//   S^T = K Q^T   : A = K fragment (row = key), B = Q fragment (col = query, held in registers).  The K ROW fed to
//                   MFMA row i of 16-key tile kt is key 32(kt>>1) + 8(i>>2) + 4(kt&1) + (i&3), so lane
//                   (q = lane&15, g = lane>>4), which receives rows 4g..4g+3 of every tile, owns the 8 CONTIGUOUS
//                   keys 32kk + 8g .. +7 of each 32-key step kk (tiles 2kk and 2kk+1).
//   softmax       : per-lane max guard and exponentials only; no cross-lane step and no LDS round trip in the hot path
//                   (the cross-lane row maximum is needed only inside the rare re-centring branch).  bf16: no maximum at all
//                   after tile 0 (template flag NOMAX: overflow check + tracking second pass).
//   l^T += 1 P^T  : row sums accumulate on the matrix pipe (all-ones A fragment; template flag LS, default).
//   O^T += V^T P^T: A = V^T fragment (row = d, 8 contiguous keys 32kk + 8g..+7: one ds_read_b128),
//                   B = P^T packed in registers straight from the S^T accumulators (no transpose, no LDS);
//                   V^T rows are key-contiguous because the QKV GEMM epilogue wrote V transposed ([B][H][64][Npad]).
//   K and V^T tiles (64 keys) are double-buffered in LDS via global_load_lds_dwordx4 (rz_common.h panels).
#include <type_traits>

#include "rz_common.h"
#include "rz_kernels.h"
#ifdef RZ_EXPERIMENTS
#include "attn_ks_loop.inc"
#endif

namespace rz {

typedef float fa_f32x32 __attribute__((ext_vector_type(32)));
typedef unsigned fa_u32x32 __attribute__((ext_vector_type(32)));
typedef unsigned fa_u32x4 __attribute__((ext_vector_type(4)));

constexpr int FA_QROWS = 128;   // query rows per 4-wave workgroup (the 8-wave variant covers 256)
constexpr int FA_KEYS = 64;     // keys per KV tile
#ifndef RZ_FA_WAVES
#define RZ_FA_WAVES 3
#endif
constexpr float FA_DEFER = 8.0f;  // a row is re-centred only when its max grew by more than 2^8 (P <= 256: safe in fp32/bf16/f16)

template <typename T> struct FaCfg {
    static constexpr int NPAN = 64 * (int)sizeof(T) / 128;   // 128-B panels per 64-element row (1 or 2)
    static constexpr int TILE_BYTES = NPAN * 64 * 128;       // one K (or V^T) tile
};

// 4 elements at `key_a` and 4 at `key_b` of row `row` of a [64 rows][64 elems] tile stored as NPAN panels.
template <typename T>
__device__ __forceinline__ typename Traits<T>::frag lds_frag_split(const char* tile, int row, int key_a, int key_b) {
    typedef typename Traits<T>::vec4 v4;
    auto addr = [&](int key) {
        const int byte = key * (int)sizeof(T);
        const int pan = byte >> 7, within = byte & 127;
        return tile + pan * (64 * 128) + panel_off(row, within >> 4) + (within & 15);
    };
    const v4 lo = *reinterpret_cast<const v4*>(addr(key_a));
    const v4 hi = *reinterpret_cast<const v4*>(addr(key_b));
    typename Traits<T>::frag f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// fragment of 8 consecutive elements starting at element `e0` (multiple of 8) of row `row`
template <typename T>
__device__ __forceinline__ typename Traits<T>::frag lds_frag_row(const char* tile, int row, int e0) {
    const int byte = e0 * (int)sizeof(T);
    const int pan = byte >> 7;
    const int kb = (byte & 127) / (8 * (int)sizeof(T));
    return lds_frag<T>(tile + pan * (64 * 128), row, kb);
}

// NW = waves per workgroup (4 or 8): 8 waves share each K/V tile -> half the L2->LDS staging bytes per FLOP
// (one CU sustains only ~50 GB/s of global_load_lds traffic, tools/mb_ldsdma.hip), at <=128 VGPRs for 2 WGs/CU.
// QT = 16-row query tiles per wave (2 or 4): QT = 4 halves both the K/V staging bytes and the LDS fragment reads per MFMA
// (every K / V^T fragment feeds 4 MFMAs instead of 2) at the price of ~250 VGPRs (2 waves per SIMD).
// LS = row sums on the matrix pipe: l^T += 1 P^T with an all-ones A fragment (two more MFMAs per 16-row query tile and KV
// tile, every lane then holds l of its own query in all four registers) instead of 32 v_add_f32 per wave and tile — the
// kernel is bound by VALU / MFMA ISSUE slots (VALU active 67 %, MFMA busy 45 %, issue-stalled 45 % of wave cycles,
// profiles/r01/mfma_utilisation_pmc.json), and the matrix pipe has room.  l then sums the T-rounded P that also feeds PV.
// NOMAX = no running maximum in the hot loop (bf16 / f32 operands): softmax is scale free in floating point, so the reference point
// fixed by tile 0 (its true row maximum) stays valid until some 2^(s - m) or a sum of them leaves the f32 / bf16 range, i.e. a later
// score exceeds tile 0's maximum by > ~100 in log2 units.  That cannot be ruled out, so the kernel checks l and O for inf / nan when
// the loop is done and, if any row of the workgroup overflowed, runs the whole loop again with the tracking path (16 v_max3 +
// the vote per tile come out of the hot loop: the kernel is bound by vector ISSUE slots).  f16 operands keep the tracking path: P <= 65504 means the
// second pass already triggers at s - m > 16, which random scores reach often enough to cost more than the maxima (927 against 997 TFLOP/s).
// ABL (only instantiated != 0 with -DRZ_EXPERIMENTS, attn_variant 1000 + ABL): TIMING ablations of the hot loop, results wrong by
// construction (tools/attn_ablate.py): 1 no exponentials, 2 no P V / row-sum MFMAs, 4 no score MFMAs,
// 8 no LDS fragment reads, 16 no K / V staging after tile 1, 32 no barriers, 64 a quarter of the LDS fragment reads.
// NBUF = K / V^T tiles resident in LDS (2: tile t+1 is staged while tile t is consumed; 3: tile t+2 is — two tiles of time for the
// LDS-DMA to land, waited for with a counted vmcnt).
template <typename T, int NW, int QT, bool LS = false, bool NOMAX = false, int ABL = 0, int NBUF = 2>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : ((QT == 4 || sizeof(T) == 4) ? 2 : RZ_FA_WAVES)) void flash_attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ vT, T* __restrict__ ctx,
                                                            int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad, const unsigned* __restrict__ run_if) {
    typedef typename Traits<T>::frag frag_t;
    if constexpr (sizeof(T) == 4) {      // exact-fp32 operands: predicated launch (fp32 mode's overflow guard, rz_kernels.h GemmArgs::run_if)
        if (run_if && *run_if == 0) return;
    }
    constexpr int NPAN = FaCfg<T>::NPAN;
    constexpr int TILE = FaCfg<T>::TILE_BYTES;
    constexpr int ES = (int)sizeof(T);
    __shared__ __attribute__((aligned(1024))) char lds[2 * NBUF * TILE];   // K0 .. K(NBUF-1) V0 .. V(NBUF-1)
    __shared__ int overflowed;                                      // NOMAX: some row of this workgroup left the float range

    const int tid = threadIdx.x, lane = tid & 63;
    if (NOMAX && tid == 0) overflowed = 0;                          // ordered before its readers by the loop's barriers
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably uniform: LDS-DMA bases go to M0 by SALU only
    const int l15 = lane & 15, lg = lane >> 4;

    // XCD-aware mapping: the (image, head, query block) work items, pair-major, are cut into 8 equal contiguous ranges, one per
    // XCD (blocks b and b + 8 share an XCD under round-robin dispatch): the query blocks of one (image, head) pair run on one XCD
    // — two where a range boundary falls inside the pair — so its K / V stay in that L2, and every XCD gets the same number of
    // items whatever B * H is (dealing whole pairs left XCDs 4-7 idle half the time at B * H = 12: 720 instead of 1045 TFLOP/s).
    constexpr int QROWS = 16 * QT * NW;
    const int nq = n_pad / QROWS;
    const int pairs = B * H;
    const int items = pairs * nq;
    const int per_xcd = (items + 7) >> 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int item = xcd * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int pair = item / nq;
    const int qb = item - pair * nq;
    const int b = pair / H, h = pair % H;

    const T* qbase = q + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64;
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64);
    const char* vbase = reinterpret_cast<const char*>(vT + ((int64_t)pair * 64) * n_pad);
    const int64_t k_ld = 64 * (int64_t)ES;
    const int64_t v_ld = (int64_t)n_pad * ES;

    // Q fragments: qf[qt][ks] = Q[row q0 + qt*16 + l15][d = ks*32 + lg*8 .. +7]
    const int q0 = qb * QROWS + wave * (16 * QT);
    frag_t qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[qt][ks] = *reinterpret_cast<const frag_t*>(qbase + (int64_t)(q0 + qt * 16 + l15) * 64 + ks * 32 + lg * 8);

    // ---- per-lane LDS byte offsets, loop invariant (row blocks add immediates) ----
    // K fragment for tile kt: row = 32(kt>>1) + 4(kt&1) + [8(l15>>2) + (l15&3)], 8 elements at d = 32ks + 8lg.
    // The K tile uses the swz_k XOR (depends on l15 only for these rows): conflict-free for this row set.
    const int krow = 8 * (l15 >> 2) + (l15 & 3);
    const int ksw = swz_k(krow);
    int koff[2][2];          // [ks][half]  (half: second 16-B chunk of an f32 fragment)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int byte = (ks * 32 + lg * 8) * ES + hf * 16;
            koff[ks][hf] = (byte >> 7) * (64 * 128) + krow * 128 + ((((byte & 127) >> 4) ^ ksw) << 4);
        }
    // V^T fragment (row = d 16dt + l15): 8 contiguous keys 32kk + 8lg (standard panel XOR, depends on l15 only)
    const int sw = (l15 >> 1) & 7;
    int voff[2][2];          // [kk][half]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int byte = (kk * 32 + lg * 8) * ES + hf * 16;
            voff[kk][hf] = (byte >> 7) * (64 * 128) + l15 * 128 + ((((byte & 127) >> 4) ^ sw) << 4);
        }

    auto stage = [&](int t, int buf) {
        char* sk = lds + buf * TILE;
        char* sv = lds + (NBUF + buf) * TILE;
        const int key0 = t * FA_KEYS;
#pragma unroll
        for (int p = 0; p < NPAN; ++p) {
#pragma unroll
            for (int i = 0; i < 8 / NW; ++i) {
                const int row8 = (wave * (8 / NW) + i) * 8;
                glds_rows8<1>(sk + p * (64 * 128) + row8 * 128, kbase + (int64_t)key0 * k_ld + p * 128, k_ld, row8, lane);
                glds_rows8(sv + p * (64 * 128) + row8 * 128, vbase + (int64_t)key0 * ES + p * 128, v_ld, row8, lane);
            }
        }
    };

    f32x4 oacc[QT][4];
#pragma unroll
    for (int a = 0; a < QT; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) oacc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Scores arrive in log2 units (log2(e)/sqrt(dh) is folded into the packed q weights), so the softmax is
    // 2^(s - m).  mrow = reference point of the running softmax (>= true running max - FA_DEFER); the score
    // accumulators are INITIALISED to -mrow, so the MFMA chain leaves s - mrow and the common path is
    // max -> exp2 -> sum with no subtraction and no rescale of O (re-centre only when a row's max grew by more
    // than 2^FA_DEFER since the last re-centring; wave-uniform branch).
    float mrow[QT], lrow[QT];                 // reference point; lane-partial running sum
    f32x4 cinit[QT];                          // {-m,-m,-m,-m}: C operand of the first score MFMA of every tile (no per-tile v_mov)
    f32x4 lacc[QT];                           // LS: running row sums, accumulated by the matrix pipe
    const frag_t ones = pack8<T>(1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int a = 0; a < QT; ++a) { mrow[a] = 0.f; lrow[a] = 0.f; cinit[a] = (f32x4){0.f, 0.f, 0.f, 0.f}; lacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int ntiles = (n_valid + FA_KEYS - 1) / FA_KEYS;
    // one KV tile.  FIRST: tile 0 (establishes the reference point).  MASK: ragged last tile (keys >= n_valid dead).
    auto tile = [&](int t, auto first_c, auto mask_c, auto track_c) {
        constexpr bool FIRST = decltype(first_c)::value, MASK = decltype(mask_c)::value;
        constexpr bool TRACK = FIRST || decltype(track_c)::value;    // running maximum + re-centring in this tile
        const int buf = NBUF == 2 ? (t & 1) : t % NBUF;
        const char* sk = lds + buf * TILE;
        const char* sv = lds + (NBUF + buf) * TILE;
        // ---- S' = K Q^T - mrow ----
        f32x4 sacc[QT][4];
        auto load_k = [&](int ks, int kt) -> frag_t {
            if constexpr ((ABL & 8) != 0) { frag_t x; asm volatile("" : "=v"(x)); return x; }   // whatever the registers hold: no instruction, no CSE
            if constexpr ((ABL & 64) != 0) { if (kt != 0) { frag_t x; asm volatile("" : "=v"(x)); return x; } }   // a quarter of the reads
            const int krb = (32 * (kt >> 1) + 4 * (kt & 1)) * 128;   // immediate
            if constexpr (ES == 4) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(sk + koff[ks][0] + krb);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(sk + koff[ks][1] + krb);
                return pack8<T>(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
            } else {
                return *reinterpret_cast<const frag_t*>(sk + koff[ks][0] + krb);
            }
        };
        auto load_v = [&](int kk, int dt) -> frag_t {
            if constexpr ((ABL & 8) != 0) { frag_t x; asm volatile("" : "=v"(x)); return x; }
            if constexpr ((ABL & 64) != 0) { if (dt != 0) { frag_t x; asm volatile("" : "=v"(x)); return x; } }
            if constexpr (ES == 4) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(sv + voff[kk][0] + dt * 2048);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(sv + voff[kk][1] + dt * 2048);
                return pack8<T>(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
            } else {
                return *reinterpret_cast<const frag_t*>(sv + voff[kk][0] + dt * 2048);
            }
        };
        // all 8 K fragments are requested before the first MFMA waits (LDS latency overlaps the MFMA chain)
        frag_t kf[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) kf[ks][kt] = load_k(ks, kt);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int a = 0; a < QT; ++a) {
                if constexpr ((ABL & 4) != 0) { asm volatile("" : "=v"(sacc[a][kt]) : "v"(kf[0][kt]), "v"(kf[1][kt])); }
                else sacc[a][kt] = mma(kf[0][kt], qf[a][0], cinit[a]);     // = K Q^T - m (cinit is 0 on tile 0)
            }
        }
        if constexpr ((ABL & 4) == 0) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int a = 0; a < QT; ++a) sacc[a][kt] = mma(kf[1][kt], qf[a][1], sacc[a][kt]);
        }
        }
        // V^T fragments of the first 32-key step: requested now, consumed after the softmax
        frag_t vf0[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vf0[dt] = load_v(0, dt);
        asm volatile("" ::: "memory");
        if constexpr (MASK) {
            const int key0 = t * FA_KEYS;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dead = key0 + 32 * (kt >> 1) + 8 * lg + 4 * (kt & 1) + r >= n_valid;
                    if (dead) {
#pragma unroll
                        for (int a = 0; a < QT; ++a) sacc[a][kt][r] = -INFINITY;
                    }
                }
        }
        // ---- maxima of S': per LANE (one query, 16 of its 64 keys); the other three lanes of the query are only
        // consulted when a re-centring actually happens — "any lane above the threshold" decides the same thing as
        // "any row above the threshold", so the hot path carries no cross-lane step (the ds_bpermute pairs and their
        // lgkmcnt(0) waits used to sit between the score MFMAs and the exponentials of every tile).
        float mx[QT];
        if constexpr (TRACK) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            // plain fmaxf chains: hipcc fuses them into v_max3_f32 (this file is built with -fno-honor-nans, so no
            // canonicalising v_max per MFMA output).  NOT inline asm: an asm VALU op reading an MFMA result gets
            // none of the MFMA->VALU wait states the compiler inserts for its own instructions.
            float m0 = fmaxf(fmaxf(sacc[qt][0][0], sacc[qt][0][1]), sacc[qt][0][2]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][0][3]), sacc[qt][1][0]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][1][1]), sacc[qt][1][2]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][1][3]), sacc[qt][2][0]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][2][1]), sacc[qt][2][2]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][2][3]), sacc[qt][3][0]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][3][1]), sacc[qt][3][2]);
            mx[qt] = fmaxf(m0, sacc[qt][3][3]);
        }
        }
        auto row_max = [&](float m0) {
            m0 = fmaxf(m0, __shfl_xor(m0, 16, 64));
            return fmaxf(m0, __shfl_xor(m0, 32, 64));
        };
        // ---- re-centre ----
        if constexpr (FIRST) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                mx[qt] = row_max(mx[qt]);
                mrow[qt] = mx[qt];      // tile 0 always holds >= 1 live key: finite
                cinit[qt] = (f32x4){-mx[qt], -mx[qt], -mx[qt], -mx[qt]};
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) sacc[qt][kt] -= mx[qt];
            }
        } else if constexpr (TRACK) {
        float mtop;
        if constexpr (QT == 4) mtop = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
        else if constexpr (QT == 1) mtop = mx[0];
        else mtop = fmaxf(mx[0], mx[1]);
        if (__builtin_expect(__any(mtop > FA_DEFER), 0)) {
            asm volatile("" ::: "memory");   // keeps hipcc from if-converting the rare path into the hot one
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const float delta = fmaxf(row_max(mx[qt]), 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                mrow[qt] += delta;
                cinit[qt] -= delta;
                if constexpr (LS) lacc[qt] *= alpha; else lrow[qt] *= alpha;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) sacc[qt][kt] -= delta;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] *= alpha;
            }
            asm volatile("" ::: "memory");
        }
        }
        // ---- P = 2^S', row sums, pack P^T fragments ----
        frag_t pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float pv[4][4];
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr ((ABL & 1) != 0) pv[kt][r] = sacc[qt][kt][r]; else
                    pv[kt][r] = __builtin_amdgcn_exp2f(sacc[qt][kt][r]);
                    if constexpr (!LS) psum += pv[kt][r];
                }
            if constexpr (!LS) lrow[qt] += psum;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                pf[qt][kk] = pack8<T>(pv[2 * kk][0], pv[2 * kk][1], pv[2 * kk][2], pv[2 * kk][3],
                                      pv[2 * kk + 1][0], pv[2 * kk + 1][1], pv[2 * kk + 1][2], pv[2 * kk + 1][3]);
        }
        // ---- O^T += V^T P^T ----
        frag_t vf1[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vf1[dt] = load_v(1, dt);
        asm volatile("" ::: "memory");
        if constexpr ((ABL & 2) != 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) asm volatile("" :: "v"(vf0[dt]), "v"(vf1[dt]));
#pragma unroll
            for (int a = 0; a < QT; ++a) asm volatile("" :: "v"(pf[a][0]), "v"(pf[a][1]));
            return;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int a = 0; a < QT; ++a) oacc[a][dt] = mma(vf0[dt], pf[a][0], oacc[a][dt]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int a = 0; a < QT; ++a) oacc[a][dt] = mma(vf1[dt], pf[a][1], oacc[a][dt]);
        }
        if constexpr (LS) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < QT; ++a) lacc[a] = mma(ones, pf[a][kk], lacc[a]);
        }
    };
    using TrueT = std::integral_constant<bool, true>;
    using FalseT = std::integral_constant<bool, false>;

    const bool ragged = (n_valid % FA_KEYS) != 0;
    const int nplain = ragged ? ntiles - 1 : ntiles;     // tiles [0, nplain) need no masking
    auto run = [&](auto track_c) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ntiles > 1) stage(1, 1);
        if constexpr (NBUF == 3) {
            // three tiles resident: tile t+2 is requested at the top of tile t (its buffer was tile t-1's, which every wave left at the last
            // barrier) and only tile t+1 has to be there at the bottom -> this wave's DMA_PER_STAGE pieces of tile t+2 may stay in flight
            constexpr int DMA_PER_STAGE = NPAN * (8 / NW) * 2;
            static_assert(DMA_PER_STAGE == 4 || DMA_PER_STAGE == 8, "counted vmcnt below");
            auto bottom = [&](bool staged) {
                if (staged) { if constexpr (DMA_PER_STAGE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr ((ABL & 32) == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            };
            if constexpr ((ABL & 16) == 0) { if (ntiles > 2) stage(2, 2); }
            if (nplain >= 1) tile(0, TrueT{}, FalseT{}, track_c); else tile(0, TrueT{}, TrueT{}, track_c);
            bottom(ntiles > 2);
            for (int t = 1; t < nplain; ++t) {
                const bool st = t + 2 < ntiles;
                if constexpr ((ABL & 16) == 0) { if (st) stage(t + 2, (t + 2) % 3); }
                tile(t, FalseT{}, FalseT{}, track_c);
                bottom(st);
            }
            if (ragged && ntiles > 1) tile(ntiles - 1, FalseT{}, TrueT{}, track_c);
            return;
        }
        if (nplain >= 1) tile(0, TrueT{}, FalseT{}, track_c); else tile(0, TrueT{}, TrueT{}, track_c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 1; t < nplain; ++t) {
            if constexpr ((ABL & 16) == 0) { if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1); }
            tile(t, FalseT{}, FalseT{}, track_c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr ((ABL & 32) == 0) __syncthreads();
        }
        if (ragged && ntiles > 1) tile(ntiles - 1, FalseT{}, TrueT{}, track_c);
    };
    if constexpr (NOMAX) {
        run(FalseT{});
        // inf / nan anywhere in l or O (bit test: this file is built with -fno-honor-nans)?
        auto left_range = [](float x) { return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u; };
        bool bad = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            bad |= left_range(LS ? lacc[qt][0] : lrow[qt]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) bad |= left_range(oacc[qt][dt][r]);
        }
        if (__any(bad) && lane == 0) overflowed = 1;
        __syncthreads();                                 // also: every wave is done with the K / V buffers
        if (ABL == 0 && __builtin_expect(overflowed != 0, 0)) {      // workgroup-uniform: the waves share the staging and its barriers
#pragma unroll
            for (int a = 0; a < QT; ++a) {
                mrow[a] = 0.f; lrow[a] = 0.f; cinit[a] = (f32x4){0.f, 0.f, 0.f, 0.f}; lacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) oacc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            run(TrueT{});
        }
    } else {
        run(TrueT{});
    }

    // ---- epilogue: O = O^T / l, ctx[(b*n_pad + q)][h*64 + d] ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l;
        if constexpr (LS) {
            l = lacc[qt][0];            // every row of the ones-tile is the same sum: no cross-lane step
        } else {
            l = lrow[qt];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int qrow = q0 + qt * 16 + l15;
        T* o = ctx + ((int64_t)b * n_pad + qrow) * (H * 64) + h * 64 + lg * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 v = oacc[qt][dt] * inv;
            *reinterpret_cast<typename Traits<T>::vec4*>(o + dt * 16) = pack4<T>(v[0], v[1], v[2], v[3]);
        }
    }
}

#ifdef RZ_EXPERIMENTS
// ---------------------------------------------------------------------------------------------
// flash_attn_ks_kernel ("key split"): the LDS bytes a wave reads per FLOP depend only on the query rows it owns (it reads its whole share
// of every K / V^T tile whatever the MFMA shape), and the timing ablations of flash_attn_kernel name those reads as what holds the
// clock at 2.0 GHz.  Here a workgroup owns 256 query rows; wave (qg, kh) owns 128 of them (qg) and HALF of every 64-key tile (kh: the
// 32 keys of 32-key step kk = kh): 8 fragment reads per 72 MFMAs instead of 16 per 36.  The two key halves of a query group run as
// independent softmax streams with their own reference points and are merged once, through LDS, when the keys are exhausted
// (O = O0 2^(m0 - M) + O1 2^(m1 - M), l likewise).  ~400 registers: one wave per SIMD, one workgroup per CU.
// NOMAX as in flash_attn_kernel (reference point = the wave's maximum over ITS keys of tile 0; overflow -> tracked second pass).
// ---------------------------------------------------------------------------------------------
#ifndef RZ_ATTN_KS_LOOP_BF16_NOVALU        // the ablated texts exist only when the generator ran with RZ_KS_ABLATIONS=1
#define RZ_ATTN_KS_LOOP_BF16_NOVALU RZ_ATTN_KS_LOOP_BF16
#define RZ_ATTN_KS_LOOP_BF16_NODMA RZ_ATTN_KS_LOOP_BF16
#define RZ_ATTN_KS_LOOP_BF16_NORDS RZ_ATTN_KS_LOOP_BF16
#define RZ_ATTN_KS_LOOP_BF16_NOBAR RZ_ATTN_KS_LOOP_BF16
#define RZ_ATTN_KS_LOOP_BF16_NOP1MFMA RZ_ATTN_KS_LOOP_BF16
#define RZ_ATTN_KS_LOOP_BF16_NOPVMFMA RZ_ATTN_KS_LOOP_BF16
#define RZ_ATTN_KS_LOOP_BF16_MFMAONLY RZ_ATTN_KS_LOOP_BF16
#endif
// The same decomposition in plain C++ with the running maximum tracked in every tile: what a workgroup of flash_attn_ks_kernel falls back
// to when some 2^S left the float range (never seen on the model's data; forced by tests).  ~400 live values: hipcc spills, nobody cares.
template <typename T>
__device__ __noinline__ void flash_attn_ks_tracked(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vT, T* __restrict__ ctx,
                                                   int64_t qk_batch_stride, int H, int n_valid, int n_pad, int b, int h, int pair, int qb, char* lds) {
    typedef typename Traits<T>::frag frag_t;
    static_assert(sizeof(T) == 2, "16-bit operands only");
    constexpr int TILE = FaCfg<T>::TILE_BYTES;       // 8 KB
    constexpr int QT = 8;                            // 16-row query tiles per wave
    constexpr int XBUF = 36 * 1024;                  // hand-over per query group: 32 O fragments x 1 KB, then l and m (2 KB each)
    constexpr int QROWS = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave >> 1, kh = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const T* qbase = q + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64;
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64);
    const char* vbase = reinterpret_cast<const char*>(vT + ((int64_t)pair * 64) * n_pad);
    const int64_t k_ld = 64 * 2, v_ld = (int64_t)n_pad * 2;

    const int q0 = qb * QROWS + qg * 128;
    frag_t qf[QT][2];
#pragma unroll
    for (int a = 0; a < QT; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[a][ks] = *reinterpret_cast<const frag_t*>(qbase + (int64_t)(q0 + a * 16 + l15) * 64 + ks * 32 + lg * 8);

    // K fragment of the wave's 16-key tile kt (0, 1): row = 32 kh + 4 kt + [8 (l15 >> 2) + (l15 & 3)]; the XOR term depends on l15 only
    const int krow = 8 * (l15 >> 2) + (l15 & 3);
    const int ksw = swz_k(krow);
    int koff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) koff[ks] = (32 * kh + krow) * 128 + (((ks * 4 + lg) ^ ksw) << 4);
    // V^T fragment: row d = 16 dt + l15, the 8 contiguous keys 32 kh + 8 lg .. +7
    const int voff = l15 * 128 + (((kh * 4 + lg) ^ ((l15 >> 1) & 7)) << 4);

    auto stage = [&](int t, int buf) {
        char* sk = lds + buf * TILE;
        char* sv = lds + (2 + buf) * TILE;
        const int key0 = t * FA_KEYS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row8 = (wave * 2 + i) * 8;
            glds_rows8<1>(sk + row8 * 128, kbase + (int64_t)key0 * k_ld, k_ld, row8, lane);
            glds_rows8(sv + row8 * 128, vbase + (int64_t)key0 * 2, v_ld, row8, lane);
        }
    };

    f32x4 oacc[QT][4], lacc[QT], cinit[QT];
    float mrow[QT];
    const frag_t ones = pack8<T>(1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f);
    auto reset = [&]() {
#pragma unroll
        for (int a = 0; a < QT; ++a) {
            mrow[a] = 0.f; cinit[a] = (f32x4){0.f, 0.f, 0.f, 0.f}; lacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) oacc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    reset();
    auto row_max = [&](float m0) {
        m0 = fmaxf(m0, __shfl_xor(m0, 16, 64));
        return fmaxf(m0, __shfl_xor(m0, 32, 64));
    };

    auto tile = [&](int t, auto first_c, auto mask_c, auto track_c) {
        constexpr bool FIRST = decltype(first_c)::value, MASK = decltype(mask_c)::value;
        constexpr bool TRACK = FIRST || decltype(track_c)::value;
        const int buf = t & 1;
        const char* sk = lds + buf * TILE;
        const char* sv = lds + (2 + buf) * TILE;
        frag_t kf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kf[ks][kt] = *reinterpret_cast<const frag_t*>(sk + koff[ks] + kt * (4 * 128));
        f32x4 sacc[QT][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int a = 0; a < QT; ++a) sacc[a][kt] = mma(kf[0][kt], qf[a][0], cinit[a]);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int a = 0; a < QT; ++a) sacc[a][kt] = mma(kf[1][kt], qf[a][1], sacc[a][kt]);
        frag_t vf[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vf[dt] = *reinterpret_cast<const frag_t*>(sv + voff + dt * 2048);
        asm volatile("" ::: "memory");
        if constexpr (MASK) {
            const int key0 = t * FA_KEYS + 32 * kh + 8 * lg;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (key0 + 4 * kt + r >= n_valid) {
#pragma unroll
                        for (int a = 0; a < QT; ++a) sacc[a][kt][r] = -INFINITY;
                    }
                }
        }
        float mx[QT];
        if constexpr (TRACK) {
#pragma unroll
            for (int a = 0; a < QT; ++a) {
                float m0 = fmaxf(fmaxf(sacc[a][0][0], sacc[a][0][1]), sacc[a][0][2]);
                m0 = fmaxf(fmaxf(m0, sacc[a][0][3]), sacc[a][1][0]);
                m0 = fmaxf(fmaxf(m0, sacc[a][1][1]), sacc[a][1][2]);
                mx[a] = fmaxf(m0, sacc[a][1][3]);
            }
        }
        if constexpr (FIRST) {
#pragma unroll
            for (int a = 0; a < QT; ++a) {
                mx[a] = row_max(mx[a]);
                if (mx[a] == -INFINITY) mx[a] = -1e30f;        // this half holds no live key at all (n_valid <= 32): P = 0, and the merge ignores it
                mrow[a] = mx[a];
                cinit[a] = (f32x4){-mx[a], -mx[a], -mx[a], -mx[a]};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) sacc[a][kt] -= mx[a];
            }
        } else if constexpr (TRACK) {
            float any = fmaxf(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])), fmaxf(fmaxf(mx[4], mx[5]), fmaxf(mx[6], mx[7])));
            if (__builtin_expect(__any(any > FA_DEFER), 0)) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int a = 0; a < QT; ++a) {
                    const float delta = fmaxf(row_max(mx[a]), 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    mrow[a] += delta;
                    cinit[a] -= delta;
                    lacc[a] *= alpha;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) sacc[a][kt] -= delta;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) oacc[a][dt] *= alpha;
                }
                asm volatile("" ::: "memory");
            }
        }
        frag_t pf[QT];
#pragma unroll
        for (int a = 0; a < QT; ++a) {
            float pv[2][4];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[kt][r] = __builtin_amdgcn_exp2f(sacc[a][kt][r]);
            pf[a] = pack8<T>(pv[0][0], pv[0][1], pv[0][2], pv[0][3], pv[1][0], pv[1][1], pv[1][2], pv[1][3]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int a = 0; a < QT; ++a) oacc[a][dt] = mma(vf[dt], pf[a], oacc[a][dt]);
#pragma unroll
        for (int a = 0; a < QT; ++a) lacc[a] = mma(ones, pf[a], lacc[a]);
    };
    using TrueT = std::integral_constant<bool, true>;
    using FalseT = std::integral_constant<bool, false>;
    const int ntiles = (n_valid + FA_KEYS - 1) / FA_KEYS;
    const bool ragged = (n_valid % FA_KEYS) != 0;
    const int nplain = ragged ? ntiles - 1 : ntiles;
    auto run = [&](auto track_c) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ntiles > 1) stage(1, 1);
        if (nplain >= 1) tile(0, TrueT{}, FalseT{}, track_c); else tile(0, TrueT{}, TrueT{}, track_c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 1; t < nplain; ++t) {
            if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1);
            tile(t, FalseT{}, FalseT{}, track_c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (ragged && ntiles > 1) tile(ntiles - 1, FalseT{}, TrueT{}, track_c);
    };
    run(TrueT{});

    // ---- merge the two key halves of each query group (kh = 1 hands over, kh = 0 finishes), O = O^T / l ----
    __syncthreads();                                   // every wave is done with the K / V buffers
    char* xb = lds + qg * XBUF;
    if (kh == 1) {
#pragma unroll
        for (int a = 0; a < QT; ++a) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(xb + ((a * 4 + dt) * 64 + lane) * 16) = oacc[a][dt];
            *reinterpret_cast<float*>(xb + 32768 + (a * 64 + lane) * 4) = lacc[a][0];
            *reinterpret_cast<float*>(xb + 32768 + 2048 + (a * 64 + lane) * 4) = mrow[a];
        }
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int a = 0; a < QT; ++a) {
            const float l1 = *reinterpret_cast<const float*>(xb + 32768 + (a * 64 + lane) * 4);
            const float m1 = *reinterpret_cast<const float*>(xb + 32768 + 2048 + (a * 64 + lane) * 4);
            const float M = fmaxf(mrow[a], m1);
            const float f0 = __builtin_amdgcn_exp2f(mrow[a] - M), f1 = __builtin_amdgcn_exp2f(m1 - M);
            const float inv = 1.0f / (lacc[a][0] * f0 + l1 * f1);
            const int qrow = q0 + a * 16 + l15;
            T* o = ctx + ((int64_t)b * n_pad + qrow) * (H * 64) + h * 64 + lg * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 o1 = *reinterpret_cast<const f32x4*>(xb + ((a * 4 + dt) * 64 + lane) * 16);
                const f32x4 v = (oacc[a][dt] * f0 + o1 * f1) * inv;
                *reinterpret_cast<typename Traits<T>::vec4*>(o + dt * 16) = pack4<T>(v[0], v[1], v[2], v[3]);
            }
        }
    }
}


// The kernel: everything between the workgroup's first and last instruction that touches a tile is the generated loop
// (tools/gen_attn_ks_loop.py -> attn_ks_loop.inc; its header has the register map and the pipeline).  No reference point is subtracted
// in the loop (P = 2^S, exact while it stays inside the float range); the two key halves are then plain sums.  A workgroup whose l or O
// left the range — or whose l is 0 — recomputes its 256 rows with flash_attn_ks_tracked.
// AV: 0 the kernel; 1..7 timing ablations of the generated text (results wrong by construction, no fallback).
template <int AV>
__global__ __launch_bounds__(256, 1) void flash_attn_ks_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ vT,
                                                               bf16_t* __restrict__ ctx, int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad) {
    __shared__ __attribute__((aligned(1024))) char lds[5 * 16384];   // the loop's five tile buffers; afterwards the key halves' hand-over (4 x 17 KB); the fallback's buffers
    __shared__ int fallback;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) fallback = 0;                      // ordered before its readers by the loop's barriers
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave >> 1, kh = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    constexpr int QROWS = 256;
    const int nq = n_pad / QROWS;
    const int pairs = B * H;
    const int items = pairs * nq;
    const int per_xcd = (items + 7) >> 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int item = xcd * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int pair = item / nq;
    const int qb = item - pair * nq;
    const int b = pair / H, h = pair % H;

    const char* qbase = reinterpret_cast<const char*>(q + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64);
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64);
    const char* vbase = reinterpret_cast<const char*>(vT + ((int64_t)pair * 64) * n_pad);
    const int v_ld = n_pad * 2;
    const int q0 = qb * QROWS + qg * 128;
    const int ntiles = (n_valid + FA_KEYS - 1) / FA_KEYS;
    const bool ragged = (n_valid % FA_KEYS) != 0;

    // lane constants of the loop: fragment read offsets inside a tile buffer (K at 0, V^T at 8 KB), LDS-DMA source offsets and
    // destinations of the wave's two K and two V^T pieces per tile (rows 16 w + 8 i .. + 7; the XOR swizzles of rz_common.h)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const int krow = 8 * (l15 >> 2) + (l15 & 3);
    const int ksw = swz_k(krow);
    const unsigned koff0 = lds0 + (unsigned)((32 * kh + krow) * 128 + (((0 * 4 + lg) ^ ksw) << 4));
    const unsigned koff1 = lds0 + (unsigned)((32 * kh + krow) * 128 + (((1 * 4 + lg) ^ ksw) << 4));
    const unsigned voffa = lds0 + 8192u + (unsigned)(l15 * 128 + (((kh * 4 + lg) ^ ((l15 >> 1) & 7)) << 4));
    unsigned dk[2], dv[2], ldsk[2], ldsv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row8 = (wave * 2 + i) * 8, r = row8 + (lane >> 3);
        dk[i] = (unsigned)(r * 128 + (((lane & 7) ^ swz_k(r)) << 4));
        dv[i] = (unsigned)(r * v_ld + (((lane & 7) ^ swz_std(r)) << 4));
        ldsk[i] = lds0 + (unsigned)(row8 * 128);
        ldsv[i] = lds0 + 8192u + (unsigned)(row8 * 128);
    }
    const unsigned qoff = (unsigned)((q0 + l15) * 128 + lg * 16);
    const int vg = (n_valid - (ntiles - 1) * FA_KEYS) - (32 * kh + 8 * lg);      // live keys of the lane's 8-key group in the last tile
    const uint64_t qb64 = (uint64_t)qbase, kb64 = (uint64_t)kbase, vb64 = (uint64_t)vbase;
    fa_f32x32 o0, o1, o2, o3, ls;
#define RZ_KS_ASM(TEXT)                                                                                                                   \
    asm volatile(TEXT                                                                                                                     \
                 : "={a[0:31]}"(o0), "={a[32:63]}"(o1), "={a[64:95]}"(o2), "={a[96:127]}"(o3), "={a[128:159]}"(ls)                        \
                 : [koff0] "v"(koff0), [koff1] "v"(koff1), [voff] "v"(voffa), [dk0] "v"(dk[0]), [dk1] "v"(dk[1]), [dv0] "v"(dv[0]),        \
                   [dv1] "v"(dv[1]), [qoff] "v"(qoff), [vg] "v"(vg), [klo] "s"((unsigned)kb64), [khi] "s"((unsigned)(kb64 >> 32)),         \
                   [vlo] "s"((unsigned)vb64), [vhi] "s"((unsigned)(vb64 >> 32)), [qlo] "s"((unsigned)qb64), [qhi] "s"((unsigned)(qb64 >> 32)), \
                   [ntl] "s"((unsigned)ntiles), [n] "s"((unsigned)ntiles), [mk] "s"(ragged ? (unsigned)(ntiles - 1) : 0xffffffffu), [ldsk0] "s"(ldsk[0]),     \
                   [ldsk1] "s"(ldsk[1]), [ldsv0] "s"(ldsv[0]), [ldsv1] "s"(ldsv[1])                                                       \
                 : RZ_ATTN_KS_CLOBBERS)
    if constexpr (AV == 0) RZ_KS_ASM(RZ_ATTN_KS_LOOP_BF16);
    else if constexpr (AV == 1) RZ_KS_ASM(RZ_ATTN_KS_LOOP_BF16_NOVALU);
    else if constexpr (AV == 2) RZ_KS_ASM(RZ_ATTN_KS_LOOP_BF16_NODMA);
    else if constexpr (AV == 3) RZ_KS_ASM(RZ_ATTN_KS_LOOP_BF16_NORDS);
    else if constexpr (AV == 4) RZ_KS_ASM(RZ_ATTN_KS_LOOP_BF16_NOBAR);
    else if constexpr (AV == 5) RZ_KS_ASM(RZ_ATTN_KS_LOOP_BF16_NOP1MFMA);
    else if constexpr (AV == 6) RZ_KS_ASM(RZ_ATTN_KS_LOOP_BF16_NOPVMFMA);
    else if constexpr (AV == 8) RZ_KS_ASM(RZ_ATTN_KS_LOOP_BF16);          // the full text without the fallback (debugging)
    else RZ_KS_ASM(RZ_ATTN_KS_LOOP_BF16_MFMAONLY);
#undef RZ_KS_ASM

    // ---- the key halves are plain sums.  Wave (qg, kh) finishes the query tiles 4 kh .. 4 kh + 3 of its group and hands the other four
    // (16 O fragments + their l) to its partner through LDS ----
    __syncthreads();                                   // every wave is done with the tile buffers
    char* mine = lds + wave * 17408;
    const char* theirs = lds + (wave ^ 1) * 17408;
    auto frag_of = [&](auto a_c, int dt) -> f32x4 {    // O fragment (a, dt) out of the pinned 32-float rows: a is a compile-time constant
        constexpr int a = decltype(a_c)::value;
        const fa_f32x32& r = (a >> 1) == 0 ? o0 : (a >> 1) == 1 ? o1 : (a >> 1) == 2 ? o2 : o3;
        const int e = (a & 1) * 16 + dt * 4;
        return (f32x4){r[e], r[e + 1], r[e + 2], r[e + 3]};
    };
    auto give = [&](auto a0_c) {
        constexpr int a0 = decltype(a0_c)::value;
        auto one = [&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<f32x4*>(mine + ((i * 4 + dt) * 64 + lane) * 16) = frag_of(std::integral_constant<int, a0 + i>{}, dt);
            *reinterpret_cast<float*>(mine + 16384 + (i * 64 + lane) * 4) = ls[4 * (a0 + i)];
        };
        one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
    };
    bool bad = false;
    auto left_range = [](float x) { return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u; };      // inf / nan (built with -fno-honor-nans)
    auto keep = [&](auto a0_c) {
        constexpr int a0 = decltype(a0_c)::value;
        auto one = [&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            const float l = ls[4 * (a0 + i)] + *reinterpret_cast<const float*>(theirs + 16384 + (i * 64 + lane) * 4);
            bad |= left_range(l) || l == 0.f;
            const float inv = 1.0f / l;
            bf16_t* o = ctx + ((int64_t)b * n_pad + q0 + (a0 + i) * 16 + l15) * (H * 64) + h * 64 + lg * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 s = frag_of(std::integral_constant<int, a0 + i>{}, dt) + *reinterpret_cast<const f32x4*>(theirs + ((i * 4 + dt) * 64 + lane) * 16);
                bad |= left_range(s[0]) || left_range(s[1]) || left_range(s[2]) || left_range(s[3]);
                const f32x4 v = s * inv;
                *reinterpret_cast<Traits<bf16_t>::vec4*>(o + dt * 16) = pack4<bf16_t>(v[0], v[1], v[2], v[3]);
            }
        };
        one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
    };
    if (kh == 0) give(std::integral_constant<int, 4>{}); else give(std::integral_constant<int, 0>{});
    __syncthreads();
    if (kh == 0) keep(std::integral_constant<int, 0>{}); else keep(std::integral_constant<int, 4>{});
    if (AV == 0) {
        if (__any(bad) && lane == 0) fallback = 1;
        __syncthreads();                               // also: the hand-over buffers have been read
        if (__builtin_expect(fallback != 0, 0))
            flash_attn_ks_tracked<bf16_t>(q, k, vT, ctx, qk_batch_stride, H, n_valid, n_pad, b, h, pair, qb, lds);
    }
}

#endif
// ---------------------------------------------------------------------------------------------
// flash_attn_split_kernel: the fp32 mode's attention on the f16 matrix pipe.  Every fp32 operand x is carried as two f16 planes,
// hi = f16(x) and lo = f16(x - hi) (x - hi is exact in fp32, so hi + lo holds 22 mantissa bits), and every product as three MFMAs with
// fp32 accumulation:  a.b = ah.bh + ah.bl + al.bh  (the dropped al.bl term is 2^-22 relative).  S = K Q^T, P = 2^(S - m) in fp32,
// P split the same way, O = V^T P^T; row sums are plain fp32 adds of P.  96 f16 MFMAs per wave and 64-key tile instead of 256
// exact-fp32 ones (16x16x4_f32 runs at 1/16 of the f16 rate): 2.5 PFLOP/s / 3 against 157 TFLOP/s of matrix peak.
// Same tiles, fragment maps and XCD mapping as flash_attn_workgroup<f16, 4, 2>; LDS holds both planes of K and V^T, double buffered
// (64 KB, two workgroups per CU); the running maximum is tracked (P <= 2^8 fits f16).  Output: fp32 ctx.
// ---------------------------------------------------------------------------------------------
// OUT_SPLIT: ctx leaves as the out-projection's split A operand instead of fp32: 1 = [rows][3 * H * 64] f16 = [hi | lo | hi],
// 2 = the MX form (rz_common.h: [hi f16 x D | per head: lo8 x 64, hi8 x 64], 4 D bytes per row).
// MXA (round 4): the MX form of the two correction terms (rz_common.h "MX form"): the second plane of q, k and V^T is not f16(x - hi) but
// 128 bytes of e4m3 per row — q: [lo8 x 64 | hi8 x 64], k: [hi8 | lo8] (per key / query, along d), V^T: [hi8 | lo8] per (d, 64-key
// block), the 64 keys of a block in the order the score accumulators hand them to a lane (position 16 g + j holds key 8 g + j for j < 8,
// 32 + 8 g + j - 8 beyond: gemm_common.h writes it so) — and each pair of correction terms is ONE block-scaled MFMA over K' = 128:
// S += [k_hi8 | k_lo8] . [q_lo8 | q_hi8],  O += [v_hi8 | v_lo8] . [p_lo8 | p_hi8]  (P split in registers, fixed scales 1 and 2^-11).
// 32 f16 MFMAs + 16 block-scaled ones per wave and 64-key tile instead of 96 f16 ones; same bytes staged, same LDS image.
// MXA 1 = P V only (q and k keep their f16 lo planes: the scores stay at 22 bits — a score's error is EXPONENTIATED, and 4-bit correction
// terms leave it at 2^-16 sum |q_d k_d|, harmless on small logits only), 2 = scores as well.
// ABL (round 5, which correction terms are DROPPED): bit 0 = P V as the single product v_hi . p_hi — both operands at f16's 11 bits, whose
// rounding errors are independent per key and average down over the keys a row attends to (profiles/r05/fp32_term_ablation.log); the row sums
// then run on the matrix pipe over the same rounded P (ones fragment, as flash_attn_kernel's LS), so the weights still sum to one exactly, V^T's
// second plane is neither staged nor read (48 KB of LDS instead of 64) and the loop loses the P split (8 v_fma_mix + 8 converts per 8
// probabilities) and 16 of its 80 MFMA units.  With bit 0, MXA 1 is meaningless: 0 = scores on f16 lo planes, 2 = scores on e4m3 pairs.
// Bits 1 / 2 (accuracy ablations only, -DRZ_EXPERIMENTS): 2 = P hi only with V hi + lo (MXA 0), 4 = scores as k_hi . q_hi only.
template <int OUT_SPLIT, int MXA = 0, int ABL = 0>
__global__ __launch_bounds__(256, 2) void flash_attn_split_kernel(const f16_t* __restrict__ q, const f16_t* __restrict__ k,
                                                                  const f16_t* __restrict__ vT, void* __restrict__ ctx_out,
                                                                  int64_t qk_batch_stride, int64_t qk_lo_off, int64_t v_lo_off,
                                                                  int B, int H, int n_valid, int n_pad, unsigned* ovf_flag) {
    typedef f16x8 frag_t;
    constexpr int TILE = 64 * 128;                   // one plane of one 64-key tile
    constexpr int ES = 2;
    constexpr bool PVH = (ABL & 1) != 0;             // P V = v_hi . p_hi, row sums on the matrix pipe
    constexpr bool PH = (ABL & 2) != 0;              // P hi only, V hi + lo (f16 planes)
    constexpr bool SH = (ABL & 4) != 0;              // scores k_hi . q_hi only
    static_assert(!(PVH && MXA == 1) && !(PH && (MXA != 0 || PVH)), "see ABL");
    constexpr int VPL = PVH ? 1 : 2;                 // V^T planes resident in LDS
    __shared__ __attribute__((aligned(1024))) char lds[(4 + 2 * VPL) * TILE];   // K: [buf][plane], then V^T: [buf][plane]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;

    const int nq = n_pad / FA_QROWS;
    const int pairs = B * H;
    const int items = pairs * nq;
    const int per_xcd = (items + 7) >> 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int item = xcd * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int pair = item / nq;
    const int qb = item - pair * nq;
    const int b = pair / H, h = pair % H;

    const f16_t* qbase = q + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64;
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64);
    const char* vbase = reinterpret_cast<const char*>(vT + ((int64_t)pair * 64) * n_pad);
    const int64_t k_ld = 64 * (int64_t)ES, v_ld = (int64_t)n_pad * ES;
    const int64_t k_lo_b = qk_lo_off * ES, v_lo_b = v_lo_off * ES;

    const int q0 = qb * FA_QROWS + wave * 32;
    frag_t qh[2][2], ql[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16_t* src = qbase + (int64_t)(q0 + qt * 16 + l15) * 64 + ks * 32 + lg * 8;
            qh[qt][ks] = *reinterpret_cast<const frag_t*>(src);
            ql[qt][ks] = *reinterpret_cast<const frag_t*>(src + qk_lo_off);
        }

    const int krow = 8 * (l15 >> 2) + (l15 & 3);
    const int ksw = swz_k(krow);
    int koff[2], voff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) koff[ks] = krow * 128 + (((ks * 4 + lg) ^ ksw) << 4);
    const int sw = (l15 >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) voff[kk] = l15 * 128 + (((kk * 4 + lg) ^ sw) << 4);

    // staging: a wave's 8 loads per tile as (uniform 64-bit base of the tile) + (32-bit lane offset, fixed for the whole kernel): the saddr form of
    // global_load_lds, no per-tile 64-bit vector address arithmetic
    uint32_t kofs[2], vofs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);
        kofs[i] = (uint32_t)(r * 128 + (((lane & 7) ^ swz_k(r)) << 4));
        vofs[i] = (uint32_t)(r * (int)v_ld + (((lane & 7) ^ swz_std(r)) << 4));
    }
    auto stage = [&](int t, int buf) {
        char* sk = lds + buf * 2 * TILE;
        char* sv = lds + 4 * TILE + buf * VPL * TILE;
        const char* kt = kbase + (int64_t)t * (FA_KEYS * k_ld);
        const char* vt = vbase + (int64_t)t * (FA_KEYS * ES);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row8 = (wave * 2 + i) * 8;
                const char* kp = kt + pl * k_lo_b;           // scalar; opaque so that the plane offset is not folded into a 64-bit vector offset
                const char* vp = vt + pl * v_lo_b;
                asm("" : "+s"(kp));
                asm("" : "+s"(vp));
                if (!(SH && pl == 1))
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kp + kofs[i]),
                                                     (__attribute__((address_space(3))) void*)(sk + pl * TILE + row8 * 128), 16, 0, 0);
                if (pl < VPL)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vp + vofs[i]),
                                                     (__attribute__((address_space(3))) void*)(sv + pl * TILE + row8 * 128), 16, 0, 0);
            }
    };

    f32x4 oacc[2][4], cinit[2];
    float mrow[2];
    f32x2 lrow2[2];                  // two lane-partial running sums per query row (packed adds)
    [[maybe_unused]] f32x4 lacc[2];  // PVH: running row sums of the f16-rounded P, accumulated by the matrix pipe
    [[maybe_unused]] const frag_t ones = pack8<f16_t>(1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        mrow[a] = 0.f; lrow2[a] = (f32x2){0.f, 0.f}; lacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        cinit[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) oacc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // MXA: E8M0 scale byte of this lane's 32-element block (block = lane >> 4): [hi8 | lo8] rows (k, V^T), [lo8 | hi8] rows (q), P
    [[maybe_unused]] const int sc_hl = lg < 2 ? MX_E8_A_HI : MX_E8_A_LO, sc_lh = lg < 2 ? MX_E8_A_LO : MX_E8_A_HI, sc_p = lg < 2 ? 116 : 127;

    auto tile = [&](int t, auto first_c, auto mask_c) {
        constexpr bool FIRST = decltype(first_c)::value, MASK = decltype(mask_c)::value;
        const int buf = t & 1;
        const char* sk = lds + buf * 2 * TILE;
        const char* sv = lds + 4 * TILE + buf * VPL * TILE;
        auto load_k = [&](int pl, int ks, int kt) -> frag_t {
            return *reinterpret_cast<const frag_t*>(sk + pl * TILE + koff[ks] + (32 * (kt >> 1) + 4 * (kt & 1)) * 128);
        };
        auto load_v = [&](int pl, int kk, int dt) -> frag_t {
            return *reinterpret_cast<const frag_t*>(sv + pl * TILE + voff[kk] + dt * 2048);
        };
        f32x4 sacc[2][4];
        {
            frag_t kf[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) kf[ks][kt] = load_k(0, ks, kt);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int a = 0; a < 2; ++a) sacc[a][kt] = mma(kf[0][kt], qh[a][0], cinit[a]);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int a = 0; a < 2; ++a) sacc[a][kt] = mma(kf[1][kt], qh[a][1], sacc[a][kt]);
            if constexpr (SH) {
            } else if constexpr (MXA == 2) {
                // the key's pair row [k_hi8 | k_lo8] against the query's [q_lo8 | q_hi8] (ql[a][0 / 1] hold its chunks lg and 4 + lg)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) kf[ks][kt] = load_k(1, ks, kt);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int a = 0; a < 2; ++a) sacc[a][kt] = mma_mx(kf[0][kt], kf[1][kt], ql[a][0], ql[a][1], sacc[a][kt], sc_hl, sc_lh);
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int a = 0; a < 2; ++a) sacc[a][kt] = mma(kf[ks][kt], ql[a][ks], sacc[a][kt]);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) kf[ks][kt] = load_k(1, ks, kt);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int a = 0; a < 2; ++a) sacc[a][kt] = mma(kf[ks][kt], qh[a][ks], sacc[a][kt]);
            }
        }
        if constexpr (MASK) {
            const int key0 = t * FA_KEYS;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dead = key0 + 32 * (kt >> 1) + 8 * lg + 4 * (kt & 1) + r >= n_valid;
                    if (dead) { sacc[0][kt][r] = -INFINITY; sacc[1][kt][r] = -INFINITY; }
                }
        }
        float mx[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {       // max(max(a, b), c) chains: v_max3_f32 (as flash_attn_workgroup)
            float m0 = fmaxf(fmaxf(sacc[qt][0][0], sacc[qt][0][1]), sacc[qt][0][2]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][0][3]), sacc[qt][1][0]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][1][1]), sacc[qt][1][2]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][1][3]), sacc[qt][2][0]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][2][1]), sacc[qt][2][2]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][2][3]), sacc[qt][3][0]);
            m0 = fmaxf(fmaxf(m0, sacc[qt][3][1]), sacc[qt][3][2]);
            mx[qt] = fmaxf(m0, sacc[qt][3][3]);
        }
        auto row_max = [&](float m0) {
            m0 = fmaxf(m0, __shfl_xor(m0, 16, 64));
            return fmaxf(m0, __shfl_xor(m0, 32, 64));
        };
        if constexpr (FIRST) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                mx[qt] = row_max(mx[qt]);
                mrow[qt] = mx[qt];
                cinit[qt] = (f32x4){-mx[qt], -mx[qt], -mx[qt], -mx[qt]};
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) sacc[qt][kt] -= mx[qt];
            }
        } else if (__builtin_expect(__any(fmaxf(mx[0], mx[1]) > FA_DEFER), 0)) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float delta = fmaxf(row_max(mx[qt]), 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                mrow[qt] += delta;
                cinit[qt] -= delta;
                if constexpr (PVH) lacc[qt] *= alpha; else lrow2[qt] *= alpha;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) sacc[qt][kt] -= delta;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] *= alpha;
            }
            asm volatile("" ::: "memory");
        }
        // ---- P = 2^S' (fp32), row sums, split into planes ----
        // VALU budget (the loop is bound by the SIMD's vector issue port, profiles/NOTEBOOK.md round 4): per 8 probabilities 8 v_exp_f32, 8 v_add_f32
        // (row sums, two lane-partial sums per row), 4 v_cvt_pk_f16_f32, 8 v_fma_mix_f32 (p - f16(p) straight from the packed halves: the -1 is kept
        // opaque so that hipcc does not turn the fma back into convert + subtract) and, MXA, 4 + 4 fp8 converts, the 2^11 of the lo plane folded
        // into v_cvt_scalef32_pk_fp8_f32's scale operand.
        [[maybe_unused]] float m1 = -1.0f;
        asm("" : "+s"(m1));
        frag_t ph[2][2];
        [[maybe_unused]] frag_t pl[2][2];       // MXA: pl[qt][0] = [p_lo8 of the lane's 16 keys], pl[qt][1] = [p_hi8 ...] (16 bytes each)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            [[maybe_unused]] uint32_t w8[2][4];           // MXA: [lo8 | hi8][kk * 2 + half]
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float pv[8];
                [[maybe_unused]] float pr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pv[i] = __builtin_amdgcn_exp2f(sacc[qt][2 * kk + (i >> 2)][i & 3]);
#pragma unroll
                // plain adds into two lane-partial sums: beside MFMAs a v_pk_add_f32 costs ~3 issue slots (MI355X_MICROARCH.md constants table; measured
                // here: 94.2 -> 88.5 ms of attention per fp32 step); the empty asm keeps hipcc from SLP-packing them back
                for (int i = 0; i < 4; ++i) { if constexpr (!PVH) { lrow2[qt][0] += pv[2 * i]; lrow2[qt][1] += pv[2 * i + 1]; } }
                if constexpr (!PVH) asm volatile("" : "+v"(lrow2[qt][0]), "+v"(lrow2[qt][1]));
                ph[qt][kk] = pack8<f16_t>(pv[0], pv[1], pv[2], pv[3], pv[4], pv[5], pv[6], pv[7]);
                if constexpr (!(PVH || PH)) {                // the second plane of P
#pragma unroll
                    for (int i = 0; i < 8; ++i) pr[i] = __builtin_fmaf((float)ph[qt][kk][i], m1, pv[i]);
                    if constexpr (MXA) {
                        // P <= 2^8 (FA_DEFER): hi8 = e4m3(p) with scale 1, lo8 = e4m3((p - f16(p)) 2^11) with scale 2^-11: both within +-448
                        w8[0][kk * 2 + 0] = cvt4_e4m3_scaled(pr[0], pr[1], pr[2], pr[3], MX_CVT_SCALE_2P11);
                        w8[0][kk * 2 + 1] = cvt4_e4m3_scaled(pr[4], pr[5], pr[6], pr[7], MX_CVT_SCALE_2P11);
                        w8[1][kk * 2 + 0] = cvt4_e4m3(pv[0], pv[1], pv[2], pv[3]);
                        w8[1][kk * 2 + 1] = cvt4_e4m3(pv[4], pv[5], pv[6], pv[7]);
                    } else {
                        pl[qt][kk] = pack8<f16_t>(pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7]);
                    }
                }
            }
            if constexpr (MXA != 0 && !PVH && !PH) {
                pl[qt][0] = __builtin_bit_cast(frag_t, (u32x4){w8[0][0], w8[0][1], w8[0][2], w8[0][3]});
                pl[qt][1] = __builtin_bit_cast(frag_t, (u32x4){w8[1][0], w8[1][1], w8[1][2], w8[1][3]});
            }
        }
        // ---- O^T += V^T P^T:  vh.ph + vh.pl + vl.ph ----
        if constexpr (PVH) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                frag_t vf[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vf[dt] = load_v(0, kk, dt);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int a = 0; a < 2; ++a) oacc[a][dt] = mma(vf[dt], ph[a][kk], oacc[a][dt]);
#pragma unroll
                for (int a = 0; a < 2; ++a) lacc[a] = mma(ones, ph[a][kk], lacc[a]);
            }
        } else if constexpr (PH) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int pln = 0; pln < 2; ++pln) {
                    frag_t vf[4];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) vf[dt] = load_v(pln, kk, dt);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int a = 0; a < 2; ++a) oacc[a][dt] = mma(vf[dt], ph[a][kk], oacc[a][dt]);
                }
        } else if constexpr (MXA) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                frag_t vf[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vf[dt] = load_v(0, kk, dt);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int a = 0; a < 2; ++a) oacc[a][dt] = mma(vf[dt], ph[a][kk], oacc[a][dt]);
            }
            frag_t v8[2][4];             // the d row's pair block [v_hi8 | v_lo8], keys in the lanes' order: chunks lg and 4 + lg
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) v8[kk][dt] = load_v(1, kk, dt);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int a = 0; a < 2; ++a) oacc[a][dt] = mma_mx(v8[0][dt], v8[1][dt], pl[a][0], pl[a][1], oacc[a][dt], sc_hl, sc_p);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                frag_t vf[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vf[dt] = load_v(0, kk, dt);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        oacc[a][dt] = mma(vf[dt], pl[a][kk], oacc[a][dt]);
                        oacc[a][dt] = mma(vf[dt], ph[a][kk], oacc[a][dt]);
                    }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vf[dt] = load_v(1, kk, dt);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int a = 0; a < 2; ++a) oacc[a][dt] = mma(vf[dt], ph[a][kk], oacc[a][dt]);
            }
        }
    };
    using TrueT = std::integral_constant<bool, true>;
    using FalseT = std::integral_constant<bool, false>;

    const int ntiles = (n_valid + FA_KEYS - 1) / FA_KEYS;
    const bool ragged = (n_valid % FA_KEYS) != 0;
    const int nplain = ragged ? ntiles - 1 : ntiles;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ntiles > 1) stage(1, 1);
    if (nplain >= 1) tile(0, TrueT{}, FalseT{}); else tile(0, TrueT{}, TrueT{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 1; t < nplain; ++t) {
        if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1);
        tile(t, FalseT{}, FalseT{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (ragged && ntiles > 1) tile(ntiles - 1, FalseT{}, TrueT{});

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l;
        if constexpr (PVH) {
            l = lacc[qt][0];            // every row of the ones tile holds the same sum: no cross-lane step
        } else {
            l = lrow2[qt][0] + lrow2[qt][1];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int qrow = q0 + qt * 16 + l15;
        const int D = H * 64;
        if constexpr (OUT_SPLIT == 2) {
            char* o = reinterpret_cast<char*>(ctx_out) + ((int64_t)b * n_pad + qrow) * (4 * D);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f16x4 hi;
                uint32_t lo8, hi8;
                split4_mx(oacc[qt][dt] * inv, hi, lo8, hi8, MX_A_HI_SCALE, MX_A_LO_SCALE, ovf_flag);
                const int k = h * 64 + lg * 4 + dt * 16;
                *reinterpret_cast<f16x4*>(o + 2 * k) = hi;
                char* pr = o + mx_pair_off(D, k);
                *reinterpret_cast<uint32_t*>(pr) = lo8;
                *reinterpret_cast<uint32_t*>(pr + 64) = hi8;
            }
        } else if constexpr (OUT_SPLIT == 1) {
            f16_t* o = reinterpret_cast<f16_t*>(ctx_out) + ((int64_t)b * n_pad + qrow) * (3 * D) + h * 64 + lg * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f16x4 hi, lo;
                split4(oacc[qt][dt] * inv, hi, lo, ovf_flag);
                *reinterpret_cast<f16x4*>(o + dt * 16) = hi;
                *reinterpret_cast<f16x4*>(o + D + dt * 16) = lo;
                *reinterpret_cast<f16x4*>(o + 2 * D + dt * 16) = hi;
            }
        } else {
            float* o = reinterpret_cast<float*>(ctx_out) + ((int64_t)b * n_pad + qrow) * D + h * 64 + lg * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(o + dt * 16) = oacc[qt][dt] * inv;
        }
    }
}

// x (fp32, B slabs of `per` contiguous elements `src_stride` apart) -> compact planes hi = f16(x), lo = f16(x - hi)   (per % 4 == 0)
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, int64_t src_stride, f16_t* __restrict__ hi,
                                                        f16_t* __restrict__ lo, int64_t per4, unsigned* ovf_flag) {
    const float* src = x + (int64_t)blockIdx.y * src_stride;
    const int64_t dst0 = (int64_t)blockIdx.y * per4 * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
        f16x4 h, l;
        split4(v, h, l, ovf_flag);
        *reinterpret_cast<f16x4*>(hi + dst0 + 4 * i) = h;
        *reinterpret_cast<f16x4*>(lo + dst0 + 4 * i) = l;
    }
}

// x (fp32) -> hi = f16(x) plane + e4m3 pair plane of the MX attention (see flash_attn_split_kernel "MXA"): kind 0 = q rows [lo8 | hi8] per 64
// elements, 1 = k rows [hi8 | lo8], 2 = V^T rows (row_len tokens per row): [hi8 | lo8] per 64-token block, tokens in the lanes' order
__global__ __launch_bounds__(256) void split_mxa_kernel(const float* __restrict__ x, int64_t src_stride, f16_t* __restrict__ hi, char* __restrict__ pair,
                                                        int64_t per4, int kind, int row_len, unsigned* ovf_flag) {
    const float* src = x + (int64_t)blockIdx.y * src_stride;
    const int64_t dst0 = (int64_t)blockIdx.y * per4 * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
        f16x4 h;
        uint32_t lo8, hi8;
        split4_mx(v, h, lo8, hi8, MX_A_HI_SCALE, MX_A_LO_SCALE, ovf_flag);
        const int64_t e = dst0 + 4 * i;                   // element index; 64 elements = one 128-byte pair block
        *reinterpret_cast<f16x4*>(hi + e) = h;
        const int in64 = (int)(e & 63);
        int pos = in64;
        if (kind == 2) { (void)row_len; pos = 16 * ((in64 & 31) >> 3) + 8 * (in64 >> 5) + (in64 & 7); }        // rows are multiples of 64 tokens long
        char* blk = pair + (e >> 6) * 128;
        *reinterpret_cast<uint32_t*>(blk + pos) = kind == 0 ? lo8 : hi8;
        *reinterpret_cast<uint32_t*>(blk + 64 + pos) = kind == 0 ? hi8 : lo8;
    }
}

size_t flash_attn_split_workspace_bytes(int B, int H, int n_pad) { return (size_t)3 * B * H * n_pad * 64 * 4; }

// fp32 q, k ([B][H][n_pad][64] each, batch stride qk_batch_stride) and V^T ([B][H][64][n_pad]) -> fp32 ctx through the split kernel.
// split_ws (flash_attn_split_workspace_bytes): planes q_hi q_lo k_hi k_lo v_hi v_lo, each B*H*n_pad*64 f16.
hipError_t launch_flash_attn_f32_split(const float* q, const float* k, const float* vT, float* ctx, void* split_ws, int64_t qk_batch_stride,
                                       int B, int H, int n_valid, int n_pad, unsigned* ovf_flag, hipStream_t s, int mxa, int pv_hi) {
    if (n_pad % FA_QROWS || n_valid <= 0 || n_valid > n_pad || B <= 0 || H <= 0 || !split_ws) return hipErrorInvalidValue;
    const int64_t per = (int64_t)H * n_pad * 64, n = (int64_t)B * per;
    f16_t* q_hi = reinterpret_cast<f16_t*>(split_ws);
    f16_t* k_hi = q_hi + 2 * n;
    f16_t* v_hi = q_hi + 4 * n;
    const dim3 sgrid(256, B);
    if (pv_hi && mxa == 1) mxa = 0;               // P V on the hi planes alone: only the scores' form is left to choose
    if (mxa && n_pad % 64) return hipErrorInvalidValue;
    if (mxa == 2) {
        hipLaunchKernelGGL(split_mxa_kernel, sgrid, dim3(256), 0, s, q, qk_batch_stride, q_hi, (char*)(q_hi + n), per / 4, 0, 64, ovf_flag);
        hipLaunchKernelGGL(split_mxa_kernel, sgrid, dim3(256), 0, s, k, qk_batch_stride, k_hi, (char*)(k_hi + n), per / 4, 1, 64, ovf_flag);
    } else {
        hipLaunchKernelGGL(split_f16_kernel, sgrid, dim3(256), 0, s, q, qk_batch_stride, q_hi, q_hi + n, per / 4, ovf_flag);
        hipLaunchKernelGGL(split_f16_kernel, sgrid, dim3(256), 0, s, k, qk_batch_stride, k_hi, k_hi + n, per / 4, ovf_flag);
    }
    if (mxa && !pv_hi) hipLaunchKernelGGL(split_mxa_kernel, sgrid, dim3(256), 0, s, vT, per, v_hi, (char*)(v_hi + n), per / 4, 2, n_pad, ovf_flag);
    else hipLaunchKernelGGL(split_f16_kernel, sgrid, dim3(256), 0, s, vT, per, v_hi, v_hi + n, per / 4, ovf_flag);
    const int nq = n_pad / FA_QROWS;
    dim3 grid(((B * H * nq + 7) / 8) * 8), block(256);
#define RZ_FAS0(MX, AB) hipLaunchKernelGGL((flash_attn_split_kernel<0, MX, AB>), grid, block, 0, s, q_hi, k_hi, v_hi, (void*)ctx, per, n, n, B, H, n_valid, n_pad, ovf_flag)
    switch (mxa * 10 + (pv_hi ? 1 : 0)) {
        case 0: RZ_FAS0(0, 0); break;
        case 1: RZ_FAS0(0, 1); break;
        case 10: RZ_FAS0(1, 0); break;
        case 20: RZ_FAS0(2, 0); break;
        case 21: RZ_FAS0(2, 1); break;
        default: return hipErrorInvalidValue;
    }
#undef RZ_FAS0
    return hipGetLastError();
}

// The same kernel on operands that are ALREADY hi/lo planes (written by the split q|k / V^T epilogues, gemm_common.h): q_hi, k_hi
// with batch stride qk_batch_stride, lo planes qk_lo_off / v_lo_off elements behind the hi planes; ctx3 = [rows][3 * H * 64] f16.
hipError_t launch_flash_attn_split_planes(const void* q_hi, const void* k_hi, const void* v_hi, void* ctx3, int64_t qk_batch_stride,
                                          int64_t qk_lo_off, int64_t v_lo_off, int B, int H, int n_valid, int n_pad, unsigned* ovf_flag, hipStream_t s, int mx_out,
                                          int mxa, int abl) {
    if (n_pad % FA_QROWS || n_valid <= 0 || n_valid > n_pad || B <= 0 || H <= 0 || (mxa && !mx_out)) return hipErrorInvalidValue;
    if ((abl & 1) && mxa == 1) mxa = 0;           // P V on the hi planes alone: the e4m3 pair plane of V^T (if any) is not read
    const int nq = n_pad / FA_QROWS;
    dim3 grid(((B * H * nq + 7) / 8) * 8), block(256);
#define RZ_FAS(OS, MX, AB) hipLaunchKernelGGL((flash_attn_split_kernel<OS, MX, AB>), grid, block, 0, s, (const f16_t*)q_hi, (const f16_t*)k_hi, (const f16_t*)v_hi, \
                                              ctx3, qk_batch_stride, qk_lo_off, v_lo_off, B, H, n_valid, n_pad, ovf_flag)
    const int os = mx_out ? 2 : 1;
    const int key = os * 100 + mxa * 10 + abl;
    switch (key) {
        case 100: RZ_FAS(1, 0, 0); break;
        case 101: RZ_FAS(1, 0, 1); break;
        case 200: RZ_FAS(2, 0, 0); break;
        case 201: RZ_FAS(2, 0, 1); break;
        case 210: RZ_FAS(2, 1, 0); break;
        case 220: RZ_FAS(2, 2, 0); break;
        case 221: RZ_FAS(2, 2, 1); break;
#ifdef RZ_EXPERIMENTS                      // accuracy ablations (tools/fp32_term_ablation.py): three-plane producers only
        case 102: RZ_FAS(1, 0, 2); break;
        case 104: RZ_FAS(1, 0, 4); break;
        case 105: RZ_FAS(1, 0, 5); break;
#endif
        default: return hipErrorInvalidValue;
    }
#undef RZ_FAS
    return hipGetLastError();
}

hipError_t launch_flash_attn(int dtype, const void* q, const void* k, const void* vT, void* ctx,
                             int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad, int variant, hipStream_t s, const unsigned* run_if) {
    if (n_pad % FA_QROWS || n_valid <= 0 || n_valid > n_pad || B <= 0 || H <= 0) return hipErrorInvalidValue;
    // Round 6: 64 query rows per workgroup (16 per wave, QT = 1) where 128-row blocks leave most of the chip idle — one 518^2 image is 132 blocks for 256 CUs,
    // 768 resident workgroups: twice the workgroups.  Automatic for bf16 only, below 384 blocks of 128 rows: its hot loop keeps no running maximum, so a row's
    // arithmetic does not depend on which rows share its wave — bit-identical to the 128-row form (unless the per-workgroup overflow re-run strikes); the f16
    // kernel re-centres per WAVE (a wave-uniform branch on any of its rows), so its bits would depend on the form and with it on the batch.
    // attn_variant 401 / 402 force the 64- / 128-row form (tests, A/B; 401 also for f16: numerically equivalent, not bitwise).
    const bool small_blocks = dtype != DT_F32 && (variant == 401 || (dtype == DT_BF16 && variant != 402 && variant != 417 && (int64_t)B * H * (n_pad / FA_QROWS) * 2 <= 768));
    // `variant` = option attn_variant.  Every 16-bit kernel: 4 waves per workgroup, row sums on the matrix pipe (LS); bf16 without the
    // running maximum in the hot loop (NOMAX), f16 with it.
    //   0 / 4 (default)  32 query rows per wave (128 per workgroup), 3 waves per SIMD
    //   417              the default shape with the running maximum tracked in every tile (bf16's second pass made the first)
    // fp32 operands: 32 rows per wave, VALU row sums (the ones-row sums would cost 8 exact-f32 MFMAs per tile).
    // Measured and retired (profiles/NOTEBOOK.md; compiled only with -DRZ_EXPERIMENTS): 64 = 64 query rows per wave where n_pad is a
    // multiple of 256 (half the LDS reads and LDS-DMA bytes per FLOP: 4-6 % faster back to back, 1.5-2 % SLOWER inside the model's step),
    // 16 = VALU row sums for 16-bit operands, 8 = 8 waves x 32 rows, 5 / 65 = three K / V^T tiles resident (LDS-DMA two tiles ahead),
    // 128 = key-split asm kernel, 1000 + m / 2000 + m = timing ablations of the hot loop.
    int nw = 4, qt = 2;
#ifdef RZ_EXPERIMENTS
    if (dtype != DT_F32 && n_pad % 256 == 0 && (variant == 64 || variant == 65)) qt = 4;
#endif
    bool ls = dtype != DT_F32, track = variant == 417 || dtype == DT_F16;
    [[maybe_unused]] bool deep = false;
#define RZ_FA(TT, NWV, QTV, LSV, NOMAXV, ABLV, NBUFV)                                                                                      \
    hipLaunchKernelGGL((flash_attn_kernel<TT, NWV, QTV, LSV, NOMAXV, ABLV, NBUFV>), dim3(((B * H * (n_pad / (16 * QTV * NWV)) + 7) / 8) * 8), \
                       dim3(64 * NWV), 0, s, (const TT*)q, (const TT*)k, (const TT*)vT, (TT*)ctx, qk_batch_stride, B, H, n_valid, n_pad, run_if)
#ifdef RZ_EXPERIMENTS
    if (variant >= 1000 && variant < 3000) {                 // timing ablations: results wrong by construction (tools/attn_ablate.py)
        if (dtype != DT_BF16 || (variant >= 2000 && n_pad % 256)) return hipErrorInvalidValue;
#define RZ_FA_ABL(A) case 1000 + A: RZ_FA(bf16_t, 4, 2, true, true, A, 2); break;
#define RZ_FA_ABL4(A) case 2000 + A: RZ_FA(bf16_t, 4, 4, true, true, A, 2); break;
        switch (variant) {
            RZ_FA_ABL(0) RZ_FA_ABL(1) RZ_FA_ABL(2) RZ_FA_ABL(4) RZ_FA_ABL(8) RZ_FA_ABL(16) RZ_FA_ABL(32) RZ_FA_ABL(6) RZ_FA_ABL(7) RZ_FA_ABL(48)
            RZ_FA_ABL(56) RZ_FA_ABL(57) RZ_FA_ABL(9) RZ_FA_ABL(24) RZ_FA_ABL(64) RZ_FA_ABL(80)
            RZ_FA_ABL4(0) RZ_FA_ABL4(1) RZ_FA_ABL4(8) RZ_FA_ABL4(16) RZ_FA_ABL4(32) RZ_FA_ABL4(7) RZ_FA_ABL4(48) RZ_FA_ABL4(56) RZ_FA_ABL4(57) RZ_FA_ABL4(24)
            default: return hipErrorInvalidValue;
        }
#undef RZ_FA_ABL
#undef RZ_FA_ABL4
        return hipGetLastError();
    }
    if (variant >= 128 && variant < 137 && dtype == DT_BF16 && n_pad % 256 == 0) {     // key-split kernel (and its timing ablations)
        const dim3 grid(((B * H * (n_pad / 256) + 7) / 8) * 8), block(256);
#define RZ_KS(AVV) case 128 + AVV: hipLaunchKernelGGL((flash_attn_ks_kernel<AVV>), grid, block, 0, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vT, (bf16_t*)ctx, qk_batch_stride, B, H, n_valid, n_pad); break;
        switch (variant) { RZ_KS(0) RZ_KS(1) RZ_KS(2) RZ_KS(3) RZ_KS(4) RZ_KS(5) RZ_KS(6) RZ_KS(7) RZ_KS(8) }
#undef RZ_KS
        return hipGetLastError();
    }
    if (dtype != DT_F32) {
        if (variant == 16) { ls = false; qt = 2; }
        if (variant == 8 && n_pad % 256 == 0) { nw = 8; ls = false; qt = 2; }
        if (variant == 5) { deep = true; qt = 2; }
        if (variant == 65 && qt == 4) deep = true;
    }
#endif
    switch (dtype) {
        case DT_F32: RZ_FA(float, 4, 2, false, false, 0, 2); break;
        case DT_BF16:
#ifdef RZ_EXPERIMENTS
            if (deep && qt == 4) { RZ_FA(bf16_t, 4, 4, true, true, 0, 3); break; }
            if (deep) { RZ_FA(bf16_t, 4, 2, true, true, 0, 3); break; }
            if (nw == 8) { RZ_FA(bf16_t, 8, 2, false, false, 0, 2); break; }
            if (!ls) { RZ_FA(bf16_t, 4, 2, false, false, 0, 2); break; }
            if (qt == 4) { if (track) RZ_FA(bf16_t, 4, 4, true, false, 0, 2); else RZ_FA(bf16_t, 4, 4, true, true, 0, 2); break; }
#endif
            if (small_blocks) { RZ_FA(bf16_t, 4, 1, true, true, 0, 2); break; }
            if (track) RZ_FA(bf16_t, 4, 2, true, false, 0, 2); else RZ_FA(bf16_t, 4, 2, true, true, 0, 2);
            break;
        case DT_F16:
#ifdef RZ_EXPERIMENTS
            if (deep && qt == 2) { RZ_FA(f16_t, 4, 2, true, false, 0, 3); break; }
            if (nw == 8) { RZ_FA(f16_t, 8, 2, false, false, 0, 2); break; }
            if (!ls) { RZ_FA(f16_t, 4, 2, false, false, 0, 2); break; }
            if (qt == 4) { RZ_FA(f16_t, 4, 4, true, false, 0, 2); break; }
#endif
            if (small_blocks) { RZ_FA(f16_t, 4, 1, true, false, 0, 2); break; }
            RZ_FA(f16_t, 4, 2, true, false, 0, 2);
            break;
        default: return hipErrorInvalidValue;
    }
#undef RZ_FA
    (void)nw; (void)ls; (void)qt;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// MPNet attention (short sequences, run once per prompt set): one thread per (prompt, head, query).
// scores = q.k/8 (1/8 folded into the packed q weights) + bias[h][i][j] + mask_add[t][j];
// mask_add = 0 for attended keys, -FLT_MAX otherwise (HF additive mask) -> fully masked rows become
// uniform exactly like the reference.
// ---------------------------------------------------------------------------------------------
// fp32 mode: 64 context values of one (token, head) as [hi | lo | hi] f16 planes of the out-projection's A operand (rz_common.h split4 semantics)
__device__ __forceinline__ void text_ctx_planes(f16_t* row, int D, const float (&o)[64], float inv, unsigned* ovf_flag) {
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
        f16x4 hi, lo;
        split4((f32x4){o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv}, hi, lo, ovf_flag);
        *reinterpret_cast<f16x4*>(row + d) = hi;
        *reinterpret_cast<f16x4*>(row + D + d) = lo;
        *reinterpret_cast<f16x4*>(row + 2 * D + d) = hi;
    }
}

template <typename T>
__global__ void text_attn_kernel(const T* __restrict__ qkv, const float* __restrict__ bias,
                                 const int64_t* __restrict__ mask, T* __restrict__ ctx, int Tn, int L, int H, const unsigned* __restrict__ run_if,
                                 f16_t* __restrict__ planes, unsigned* ovf_flag) {
    if (run_if && *run_if == 0) return;      // predicated launch (fp32 mode's text guard)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y, t = blockIdx.z;
    if (i >= L) return;
    const int D3 = 3 * H * 64, D = H * 64;
    const T* qp = qkv + ((int64_t)t * L + i) * D3 + h * 64;
    float qv[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) qv[d] = to_f32(qp[d]);
    float o[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float* brow = bias + ((int64_t)h * L + i) * L;
    for (int jj = 0; jj < L; ++jj) {
        const T* kp = qkv + ((int64_t)t * L + jj) * D3 + D + h * 64;
        const T* vp = kp + D;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) s = fmaf(qv[d], to_f32(kp[d]), s);
        s += brow[jj];
        s += (mask[(int64_t)t * L + jj] != 0) ? 0.f : -3.4028234663852886e38f;
        const float mn = fmaxf(m, s);
        const float alpha = expf(m - mn);
        const float p = expf(s - mn);
        l = l * alpha + p;
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = fmaf(p, to_f32(vp[d]), o[d] * alpha);
        m = mn;
    }
    const float inv = 1.f / l;
    if (planes) {
        text_ctx_planes(planes + ((int64_t)t * L + i) * 3 * D + h * 64, D, o, inv, ovf_flag);
        return;
    }
    T* op = ctx + ((int64_t)t * L + i) * D + h * 64;
#pragma unroll
    for (int d = 0; d < 64; ++d) op[d] = from_f32<T>(o[d] * inv);
}

// Short prompts (L <= 32: every prompt the reference's tasks use): one 256-thread workgroup per (prompt, head) with q, k, v of that head in LDS
// (fp32) — scores by thread (i, j), row softmax by thread i, P V by thread (i, d), coalesced loads and stores.  The one-thread-per-query kernel
// above left 48 lanes of its wave idle at L = 16 and walked 2 L rows of 64 scalar loads per thread: 30-105 us per layer, the longest kernel
// of the text encoder (now ~8 us).  Same arithmetic up to summation order (plain max / exp / sum instead of the online form).
template <typename T>
__global__ __launch_bounds__(256) void text_attn_small_kernel(const T* __restrict__ qkv, const float* __restrict__ bias, const int64_t* __restrict__ mask,
                                                              T* __restrict__ ctx, int L, int H, const unsigned* __restrict__ run_if,
                                                              f16_t* __restrict__ planes, unsigned* ovf_flag) {
    constexpr int LMAX = 32, LD = 65;
    if (run_if && *run_if == 0) return;      // predicated launch (fp32 mode's text guard): workgroup-uniform, before any barrier
    __shared__ float qs[LMAX * LD], ks[LMAX * LD], vs[LMAX * 64], ss[LMAX * (LMAX + 1)];
    const int h = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const int D3 = 3 * H * 64, D = H * 64;
    for (int e = tid; e < L * 64; e += 256) {
        const int i = e >> 6, d = e & 63;
        const T* row = qkv + ((int64_t)t * L + i) * D3 + h * 64 + d;
        qs[i * LD + d] = to_f32(row[0]);
        ks[i * LD + d] = to_f32(row[D]);
        vs[i * 64 + d] = to_f32(row[2 * D]);
    }
    __syncthreads();
    for (int e = tid; e < L * L; e += 256) {
        const int i = e / L, j = e - i * L;
        float sc = 0.f;
#pragma unroll 16
        for (int d = 0; d < 64; ++d) sc = fmaf(qs[i * LD + d], ks[j * LD + d], sc);
        sc += bias[((int64_t)h * L + i) * L + j];
        sc += (mask[(int64_t)t * L + j] != 0) ? 0.f : -3.4028234663852886e38f;
        ss[i * (LMAX + 1) + j] = sc;
    }
    __syncthreads();
    if (tid < L) {
        float* row = ss + tid * (LMAX + 1);
        float m = -INFINITY;
        for (int j = 0; j < L; ++j) m = fmaxf(m, row[j]);
        float l = 0.f;
        for (int j = 0; j < L; ++j) { const float p = expf(row[j] - m); row[j] = p; l += p; }
        const float inv = 1.f / l;
        for (int j = 0; j < L; ++j) row[j] *= inv;
    }
    __syncthreads();
    for (int e = tid; e < L * 64; e += 256) {
        const int i = e >> 6, d = e & 63;
        float o = 0.f;
        for (int j = 0; j < L; ++j) o = fmaf(ss[i * (LMAX + 1) + j], vs[j * 64 + d], o);
        if (planes) {          // [hi | lo | hi] of the out-projection's A operand (element-wise: rz_common.h split4's arithmetic)
            const f16_t hi = (f16_t)o;
            f16_t* pr = planes + ((int64_t)t * L + i) * 3 * D + h * 64 + d;
            pr[0] = hi; pr[D] = (f16_t)(o - (float)hi); pr[2 * D] = hi;
            if ((__float_as_uint(o) & 0x7fffffffu) > 0x477fe000u && ovf_flag) atomicOr(ovf_flag, 1u);      // |o| > 65504 or NaN (integer compare: this file is built with -fno-honor-nans)
            continue;
        }
        ctx[((int64_t)t * L + i) * D + h * 64 + d] = from_f32<T>(o);
    }
}

hipError_t launch_text_attn(int dtype, const void* qkv, const float* rel_bias, const int64_t* attn_mask, void* ctx, int T, int L,
                            int H, hipStream_t s, const unsigned* run_if, void* planes_out, unsigned* ovf_flag) {
    if (T <= 0 || L <= 0 || (planes_out && dtype != DT_F32)) return hipErrorInvalidValue;
    f16_t* planes = (f16_t*)planes_out;
    if (L <= 32) {
        const dim3 g(H, T), b(256);
        switch (dtype) {
            case DT_F32: hipLaunchKernelGGL(text_attn_small_kernel<float>, g, b, 0, s, (const float*)qkv, rel_bias, attn_mask, (float*)ctx, L, H, run_if, planes, ovf_flag); break;
            case DT_BF16: hipLaunchKernelGGL(text_attn_small_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)qkv, rel_bias, attn_mask, (bf16_t*)ctx, L, H, run_if, nullptr, nullptr); break;
            case DT_F16: hipLaunchKernelGGL(text_attn_small_kernel<f16_t>, g, b, 0, s, (const f16_t*)qkv, rel_bias, attn_mask, (f16_t*)ctx, L, H, run_if, nullptr, nullptr); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    dim3 block(64), grid((L + 63) / 64, H, T);
    switch (dtype) {
        case DT_F32:
            hipLaunchKernelGGL(text_attn_kernel<float>, grid, block, 0, s, (const float*)qkv, rel_bias, attn_mask,
                               (float*)ctx, T, L, H, run_if, planes, ovf_flag);
            break;
        case DT_BF16:
            hipLaunchKernelGGL(text_attn_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)qkv, rel_bias, attn_mask,
                               (bf16_t*)ctx, T, L, H, run_if, nullptr, nullptr);
            break;
        case DT_F16:
            hipLaunchKernelGGL(text_attn_kernel<f16_t>, grid, block, 0, s, (const f16_t*)qkv, rel_bias, attn_mask,
                               (f16_t*)ctx, T, L, H, run_if, nullptr, nullptr);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rz
