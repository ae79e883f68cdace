// radzero_hip — VL-CABS head (the reference's own arithmetic: exp/cxr_pt/model/losses.py:71-105 and
// SimilarityLogit :187-240, final scaling exp/cxr_pt/model/modeling.py:311-328), always in fp32.
//
//   vhat  = l2norm(LN_shared(tokens))                     (losses.py:90-91, :213)      [vlcabs_kernel, in LDS only]
//   S     = qhat . vhat / tau                             (losses.py:219-221)          [vlcabs_kernel]
//   p     = softmax_N(S); agg = p . vhat                  (losses.py:222-224)          [vlcabs_kernel + finalize]
//   logit = qhat . agg/||agg||                            (losses.py:226-233)          [finalize]
// sim_op == "dot" (losses.py:214-215; RadZeroLoss's constructor default, not the released config): vhat = LN_shared(tokens) and
// q = LN_shared(text) WITHOUT the L2 normalisation, S = q . vhat / sqrt(D), and the final logit normalises both: q/||q|| . agg/||agg||.
// tau of S = exp(attn_temperature) when the checkpoint has one, else exp(loss_temperature) (losses.py:175-181); the returned
// `logits` are always divided by exp(loss_temperature) (modeling.py:322-328).
//
// Softmax over N tokens is split into 128-token chunks (online-softmax partials m, l, agg[D]) that the
// finalize kernel merges, so the token tensor is streamed once and nothing of size N x N exists.
// Also: bilinear similarity-map upsample (exp/cxr_pt/inference/segmentation_utils.py:62-70).
#include "rz_common.h"
#include "rz_kernels.h"

namespace rz {

constexpr int VC_CHUNK = 128;   // token rows per workgroup = per softmax chunk
constexpr int VC_TG = 16;       // prompts per workgroup
#ifndef RZ_VLCABS_WAVES
#define RZ_VLCABS_WAVES 4        // waves per workgroup of vlcabs_kernel (4 | 8): A/B with RZ_CXXFLAGS=-DRZ_VLCABS_WAVES=8
#endif
constexpr int VC_LDS_STRIDE = 772;      // floats; 772 mod 64 = 4 spreads the 16 rows of an A-fragment read over the banks

// Cross-lane reductions on the vector ALU's data-parallel primitives (DPP) instead of ds_bpermute: the kernel is a chain of dependent latencies, and
// the 104 LDS-crossbar shuffles per 16-token tile (3 wave sums per LayerNorm row, a 16-lane max and sum per softmax row) were ~100 cycles each.
// quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: after the four steps every lane of a 16-lane row holds the row's total, bit-identical
// in all 16 lanes (each step adds two values that both partners hold).
template <int CTRL> __device__ __forceinline__ float vc_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float vc_row16_sum(float v) {
    v += vc_dpp<0xB1>(v);
    v += vc_dpp<0x4E>(v);
    v += vc_dpp<0x141>(v);
    v += vc_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float vc_row16_max(float v) {
    v = fmaxf(v, vc_dpp<0xB1>(v));
    v = fmaxf(v, vc_dpp<0x4E>(v));
    v = fmaxf(v, vc_dpp<0x141>(v));
    v = fmaxf(v, vc_dpp<0x140>(v));
    return v;
}
// all 64 lanes: the four row totals through scalar registers, summed in a fixed order
__device__ __forceinline__ float vc_wave_sum(float v) {
    v = vc_row16_sum(v);
    const int vi = __builtin_bit_cast(int, v);          // v_readlane moves 32 bits: the builtin is typed int, a float argument would be CONVERTED
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16)),
                s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), s3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return (s0 + s1) + (s2 + s3);
}

// ---- fused scores + per-chunk online-softmax partials: ws[b][chunk][t] = {m, l, agg[768]} ----
// One workgroup = (128-token chunk, image, group of 16 prompts), walked as eight 16-token tiles.  Per tile:
//   1. every wave normalises four token rows in registers (shared LayerNorm, then L2) into an LDS tile [16][768] — the
//      normalised tokens never go to HBM (the former version wrote 524 MB of them at cfg 2 and read them back once per 16 prompts);
//   2. S[16 prompts][16 tokens] on the matrix pipe with the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), K = 768 split over the
//      four waves and summed through LDS; wave 0 stores S (= cos / tau) as the similarity scores;
//   3. online softmax over the tiles (running max / sum per prompt, redundantly in every wave), P^T through a wave-private
//      LDS tile into the A-operand layout;
//   4. agg[16 prompts][768] += P V on the matrix pipe, each wave owning 192 channels (12 accumulator tiles).
// Token traffic: each token row is read once per prompt group — from HBM by the first group's workgroup, from L2 / the
// Infinity Cache by the others (a 128-token chunk is 393 KB; cfg 5 reads its 36 MB of tokens 13 times, on chip).
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void vlcabs_kernel(const float* __restrict__ tokens, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, const float* __restrict__ qhat,
                                                     float inv_tau, float* __restrict__ ws, float* __restrict__ scores, int T,
                                                     int n_valid, int n_pad, int l2_tokens) {
    constexpr int D = 768, REC = D + 2;
    __shared__ __attribute__((aligned(16))) float vs[16 * VC_LDS_STRIDE];
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    constexpr int RPW = 16 / NW;             // token rows a wave normalises per tile
    constexpr int KBW = 48 / NW;             // 16-wide K blocks of the score contraction per wave (K = 768 split over the waves)
    constexpr int CPW = D / NW;              // output channels of the aggregation per wave
    constexpr int NACC = CPW / 16;           // 16 x 16 accumulator tiles per wave
    constexpr int KUNR = NW == 4 ? 4 : 3;    // score K blocks in flight (registers)
    __shared__ float sred[NW][16][17];       // [wave][token][prompt]: partial scores of the wave's K slice
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    // prompt group fastest: the workgroups that share a token chunk are dispatched together and read it once from HBM
    const int c = blockIdx.y, b = blockIdx.z, t0 = blockIdx.x * VC_TG;
    const int nchunks = gridDim.y;
    // B operand of the score MFMAs: qhat row of prompt t0 + l15 (clamped: rows >= T are computed and dropped)
    const int tq = t0 + l15;
    const float* qrow = qhat + (int64_t)(tq < T ? tq : T - 1) * D + 4 * lg;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;    // running maximum / sum of prompt l15

    for (int tile = 0; tile < VC_CHUNK / 16; ++tile) {
        const int tok0 = c * VC_CHUNK + tile * 16;
        // ---- 1. shared LayerNorm (eps 1e-5, two-pass) + L2 normalisation of 16 token rows -> LDS.  Round 5: the wave's four rows are requested
        // from HBM TOGETHER (12 x 16 B per lane in flight: one memory latency per tile instead of four back to back — the kernel is a chain of
        // latencies, MFMA busy 0.19) and their four reduction chains are independent, so they interleave; per row the arithmetic and its order are
        // unchanged (same bits).  [Round 2 measured "four rows side by side" slower at 153 VGPRs: that version also held gamma / beta per row.]
        {
            f32x4 v[RPW][3];
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int64_t row = (int64_t)b * n_pad + tok0 + wave * RPW + rr;
#pragma unroll
                for (int i = 0; i < 3; ++i) v[rr][i] = *reinterpret_cast<const f32x4*>(tokens + row * D + (lane + 64 * i) * 4);
            }
            f32x4 gm[3], be[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                gm[i] = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * 4);
                be[i] = *reinterpret_cast<const f32x4*>(beta + (lane + 64 * i) * 4);
            }
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int rl = wave * RPW + rr;
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) sm += (v[rr][i][0] + v[rr][i][1]) + (v[rr][i][2] + v[rr][i][3]);
                const float mu = vc_wave_sum(sm) * (1.0f / D);
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    v[rr][i] -= mu;
                    q += (v[rr][i][0] * v[rr][i][0] + v[rr][i][1] * v[rr][i][1]) + (v[rr][i][2] * v[rr][i][2] + v[rr][i][3] * v[rr][i][3]);
                }
                const float rstd = rsqrtf(vc_wave_sum(q) * (1.0f / D) + eps);
                float n2 = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    v[rr][i] = v[rr][i] * rstd * gm[i] + be[i];
                    n2 += (v[rr][i][0] * v[rr][i][0] + v[rr][i][1] * v[rr][i][1]) + (v[rr][i][2] * v[rr][i][2] + v[rr][i][3] * v[rr][i][3]);
                }
                const float inv = l2_tokens ? 1.0f / fmaxf(sqrtf(vc_wave_sum(n2)), 1e-12f) : 1.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(vs + rl * VC_LDS_STRIDE + (lane + 64 * i) * 4) = v[rr][i] * inv;
            }
        }
        __syncthreads();
        // ---- 2. score tile D[token 4*lg + r][prompt l15] (tokens are the MFMA rows, prompts its columns), this wave's slice of K, walked in
        // blocks of 16: lane (., lg) supplies k = 16*kb + 4*lg + u for the u-th of four MFMAs — a permuted but consistent K order for both operands.
        // Round 5: transposed against rounds 1-4 (which had prompts as rows), so that a lane owns ONE prompt and four consecutive tokens: the
        // probabilities it computes are exactly the B operand of the aggregation MFMAs (no LDS round trip of P), the softmax statistics are per-lane
        // scalars reduced over 4 registers + 2 cross-row steps (instead of a 16-lane max and sum per accumulator register), scores and partial
        // records leave as runs of four consecutive floats.
        {
            const float* arow = vs + l15 * VC_LDS_STRIDE + 4 * lg;
            f32x4 sp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll KUNR
            for (int kb = wave * KBW; kb < wave * KBW + KBW; ++kb) {
                const f32x4 tv = *reinterpret_cast<const f32x4*>(arow + kb * 16);     // token row l15
                const f32x4 qv = *reinterpret_cast<const f32x4*>(qrow + kb * 16);     // prompt row l15
#pragma unroll
                for (int u = 0; u < 4; ++u) sp = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[u], qv[u], sp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sred[wave][4 * lg + r][l15] = sp[r];                // [token][prompt]
        }
        __syncthreads();
        f32x4 sv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t4 = (sred[0][4 * lg + r][l15] + sred[1][4 * lg + r][l15]) + (sred[2][4 * lg + r][l15] + sred[3][4 * lg + r][l15]);
            if constexpr (NW == 8)
                t4 += (sred[4][4 * lg + r][l15] + sred[5][4 * lg + r][l15]) + (sred[6][4 * lg + r][l15] + sred[7][4 * lg + r][l15]);
            sv[r] = t4 * inv_tau;
        }
        const int tokb = tok0 + 4 * lg;                  // this lane's four tokens: tokb .. tokb + 3
        if (wave == 0 && tq < T) {
            float* srow = scores + ((int64_t)b * T + tq) * n_valid + tokb;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (tokb + r < n_valid) srow[r] = sv[r];
        }
        // ---- 3. online softmax over the token tiles: per lane the statistics of prompt l15 (every wave keeps the same ones)
        f32x4 e;
        {
            f32x4 sm;
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[r] = (tokb + r < n_valid) ? sv[r] : -INFINITY;
            float tm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
            tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));                                        // max over the tile's 16 tokens
            const float mn = fmaxf(m_run, tm);
            // e^x as 2^(x log2 e) on the transcendental pipe (v_exp_f32, ~1 ulp): libm's expf was ~20 instructions on this latency chain
            const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - mn) * 1.4426950408889634f);
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = (sm[r] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((sm[r] - mn) * 1.4426950408889634f);
            float es = (e[0] + e[1]) + (e[2] + e[3]);
            es += __shfl_xor(es, 16, 64);
            es += __shfl_xor(es, 32, 64);
            l_run = l_run * alpha + es;
            m_run = mn;
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] *= alpha;
        }
        // ---- 4. agg^T[channel wave*CPW + 16*i + 4*lg + r][prompt l15] += sum_token vhat[token][channel] P[prompt][token]: A = vhat^T from LDS
        // (row = channel 16*i + l15, k = lg <-> token 4*lg + ks), B = this lane's own e[ks] (k = lg, column = prompt l15)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* arow = vs + (4 * lg + ks) * VC_LDS_STRIDE + wave * CPW + l15;
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[i * 16], e[ks], acc[i], 0, 0, 0);
        }
        __syncthreads();          // vs and sred are rewritten by the next tile
    }
    if (tq < T) {
        float* rec = ws + (((int64_t)b * nchunks + c) * T + tq) * REC;
        if (wave == 0 && lg == 0) { rec[0] = m_run; rec[1] = l_run; }
        // channels wave*CPW + 16*i + 4*lg .. + 3: rec + 2 is 8-byte aligned (REC is even), not 16
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            float* o = rec + 2 + wave * CPW + i * 16 + 4 * lg;
            *reinterpret_cast<f32x2*>(o) = (f32x2){acc[i][0], acc[i][1]};
            *reinterpret_cast<f32x2*>(o + 2) = (f32x2){acc[i][2], acc[i][3]};
        }
    }
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- merge chunks, normalise, dot with qhat ----
__global__ __launch_bounds__(256) void vlcabs_finalize_kernel(const float* __restrict__ ws, const float* __restrict__ qhat,
                                                              float tau, float* __restrict__ t2i_logits,
                                                              float* __restrict__ logits, int T, int B, int nchunks, int normalize_q) {
    constexpr int D = 768, REC = D + 2;
    __shared__ float red[4];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float M = -INFINITY;
    for (int c = 0; c < nchunks; ++c) M = fmaxf(M, ws[(((int64_t)b * nchunks + c) * T + t) * REC]);
    float L = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int c = 0; c < nchunks; ++c) {
        const float* rec = ws + (((int64_t)b * nchunks + c) * T + t) * REC;
        const float w = expf(rec[0] - M);
        L += w * rec[1];
        a0 = fmaf(w, rec[2 + tid], a0);
        a1 = fmaf(w, rec[2 + tid + 256], a1);
        a2 = fmaf(w, rec[2 + tid + 512], a2);
    }
    const float invL = 1.0f / L;
    a0 *= invL; a1 *= invL; a2 *= invL;
    const float n2 = block_sum_256(a0 * a0 + a1 * a1 + a2 * a2, red);
    const float* qv = qhat + (int64_t)t * D;
    const float dot = block_sum_256(qv[tid] * a0 + qv[tid + 256] * a1 + qv[tid + 512] * a2, red);
    // sim_op "dot": the query rows are not normalised yet (losses.py:226 normalises them only here)
    const float q2 = normalize_q ? block_sum_256(qv[tid] * qv[tid] + qv[tid + 256] * qv[tid + 256] + qv[tid + 512] * qv[tid + 512], red) : 1.0f;
    if (tid == 0) {
        const float lg = dot / (fmaxf(sqrtf(n2), 1e-12f) * (normalize_q ? fmaxf(sqrtf(q2), 1e-12f) : 1.0f));
        t2i_logits[(int64_t)t * B + b] = lg;
        logits[(int64_t)b * T + t] = lg / tau;
    }
}

size_t vlcabs_workspace_floats(int B, int T, int n_pad, int D) {
    return (size_t)B * (n_pad / VC_CHUNK) * T * (D + 2);
}

hipError_t launch_vlcabs(const float* tokens, const float* ln_gamma, const float* ln_beta, float ln_eps,
                         const float* qhat, float score_denominator, float logit_tau, int sim_dot, float* ws, float* scores, float* t2i_logits,
                         float* logits, int B, int T, int n_valid, int n_pad, int D, hipStream_t s) {
    if (D != 768 || B <= 0 || T <= 0 || n_pad % VC_CHUNK || n_valid > n_pad) return hipErrorInvalidValue;
    const int nchunks = n_pad / VC_CHUNK;
#if RZ_VLCABS_WAVES == 8
    hipLaunchKernelGGL(vlcabs_kernel<8>, dim3((T + VC_TG - 1) / VC_TG, nchunks, B), dim3(512), 0, s, tokens, ln_gamma, ln_beta, ln_eps,
                       qhat, 1.0f / score_denominator, ws, scores, T, n_valid, n_pad, sim_dot ? 0 : 1);
#else
    hipLaunchKernelGGL(vlcabs_kernel<4>, dim3((T + VC_TG - 1) / VC_TG, nchunks, B), dim3(256), 0, s, tokens, ln_gamma, ln_beta, ln_eps,
                       qhat, 1.0f / score_denominator, ws, scores, T, n_valid, n_pad, sim_dot ? 0 : 1);
#endif
    hipLaunchKernelGGL(vlcabs_finalize_kernel, dim3(T, B), dim3(256), 0, s, ws, qhat, logit_tau, t2i_logits, logits, T, B, nchunks, sim_dot ? 1 : 0);
    return hipGetLastError();
}

// ---- bilinear upsample, align_corners=False (F.interpolate semantics), optional sigmoid ----
// segmentation_utils.py:62-70 (+ torch.sigmoid, attention_map_base.py:57).  HBM-WRITE bound: n_maps*H*W*4 bytes (4.29 GB at BASELINE cfg 4)
// against a source of g*g floats per map that lives in L1 / L2.  One thread produces 4 consecutive pixels (one 16-byte store) of UP_ROWS
// output rows.  Round 4: when the output is >= ~UP_ROWS x larger than the grid (every real use: 1024 / 73 = 14), the 4 x UP_ROWS outputs of a
// thread depend on at most 3 source columns x 3 source rows: those 9 values are loaded ONCE into registers and every output row is a
// register blend — 9 gather loads per 32 outputs instead of 128 (the round-3 kernel issued four dependent gathers per pixel and ran at
// 4.4 TB/s = the texture-address path, not HBM) — and the stores are non-temporal (a 4 GB write-once stream should not sweep L2 / the
// Infinity Cache).  Arithmetic per output is unchanged (same products, same order): bit-identical to the generic path, which stays as the
// per-thread fall-back for small magnifications.
#ifndef RZ_UP_ROWS
#define RZ_UP_ROWS 16
#endif
#ifndef RZ_UP_NT
#define RZ_UP_NT 1
#endif
constexpr int UP_ROWS = RZ_UP_ROWS;
// One pixel of the bilinear map with a FIXED operation order (explicit fma / mul: nothing left to the compiler's contraction choices):
// the map kernel's two paths and the fused argmax kernel must produce the same bits, or the argmax of a plateau (the clamped half-pixel
// border rows are constant) lands on a different pixel than the first maximum of the written map.
__device__ __forceinline__ float bilerp(float hy, float ly, float hx, float lx, float v00, float v01, float v10, float v11) {
    const float top = __fmaf_rn(lx, v01, __fmul_rn(hx, v00));
    const float bot = __fmaf_rn(lx, v11, __fmul_rn(hx, v10));
    return __fmaf_rn(ly, bot, __fmul_rn(hy, top));
}
// source coordinate of output index i (F.interpolate, align_corners=False: scale * (i + 0.5) - 0.5, clamped at 0), same bits everywhere
__device__ __forceinline__ float src_coord(float scale, int i, int off) {
    const float f = __fmaf_rn(scale, (float)(i + off) + 0.5f, -0.5f);
    return f < 0.f ? 0.f : f;
}
// torch.sigmoid of a map value (attention_map_base.py:57) on the transcendental pipe: 1 / (1 + 2^(-t log2 e)), v_exp_f32 + v_rcp_f32 (each
// ~1 ulp: |error| < 3e-7 absolute on a value in (0, 1)); expf() + an IEEE division made the sigmoid maps VALU-bound at 3.6 TB/s
__device__ __forceinline__ float up_sigmoid(float t) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
}
__device__ __forceinline__ float up_sel3(int k, float a, float b, float c) { return k == 0 ? a : (k == 1 ? b : c); }
__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ maps, int64_t map_stride,
                                                                float* __restrict__ out, int g, int Hout, int Wout,
                                                                float sy, float sx, int apply_sigmoid, int off_y, int off_x) {
    // (off_y, off_x): crop origin inside the virtual square map of the aspect-ratio branch (0 for the plain branch)
    const int m = blockIdx.z;
    const int ybase = blockIdx.y * UP_ROWS;
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (x4 >= Wout) return;
    const float* src = maps + (int64_t)m * map_stride;
    int x0[4], x1[4];
    float lx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x4 + i;
        const float fx = src_coord(sx, x, off_x);
        int xa = (int)fx;
        xa = xa > g - 1 ? g - 1 : xa;                         // only reachable for x >= Wout (masked below)
        x0[i] = xa;
        x1[i] = xa + (xa < g - 1 ? 1 : 0);
        lx[i] = fx - xa;
    }
    const bool full = x4 + 3 < Wout;
    const int ylast = min(ybase + UP_ROWS, Hout) - 1;
    auto ysrc = [&](int y, int& y0, int& y1, float& ly) {
        const float fy = src_coord(sy, y, off_y);
        y0 = (int)fy;
        y0 = y0 > g - 1 ? g - 1 : y0;
        y1 = y0 + (y0 < g - 1 ? 1 : 0);
        ly = fy - y0;
    };
    int ys, ys1, ye, ye1; float lt;
    ysrc(ybase, ys, ys1, lt);
    ysrc(ylast, ye, ye1, lt);
    const int xs = x0[0];
    const bool fast = (x1[3] - xs <= 2) && (ye1 - ys <= 2);
    float c[3][3];
    if (fast) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) c[a][b] = src[min(ys + a, g - 1) * g + min(xs + b, g - 1)];
    }
    auto store = [&](int y, const float (&v)[4]) {
        const int64_t base = ((int64_t)m * Hout + y) * Wout + x4;
        float* o = out + base;
        if (full && (base & 3) == 0) {
#if RZ_UP_NT
            __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(o));
#else
            *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
#endif
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x4 + i < Wout) o[i] = v[i];
        }
    };
    if (fast) {
        // the horizontal half of bilerp() — top / bot of the three candidate source rows — does not depend on the output row: once
        float hb[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float hx = 1.f - lx[i];
            const int e0 = x0[i] - xs, e1 = x1[i] - xs;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                hb[k][i] = __fmaf_rn(lx[i], up_sel3(e1, c[k][0], c[k][1], c[k][2]), __fmul_rn(hx, up_sel3(e0, c[k][0], c[k][1], c[k][2])));
        }
#pragma unroll 4
        for (int r = 0; r < UP_ROWS; ++r) {
            const int y = ybase + r;
            if (y >= Hout) break;
            int y0, y1; float ly;
            ysrc(y, y0, y1, ly);
            const float hy = 1.f - ly;
            const int d0 = y0 - ys, d1 = y1 - ys;             // wave-uniform
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = __fmaf_rn(ly, up_sel3(d1, hb[0][i], hb[1][i], hb[2][i]), __fmul_rn(hy, up_sel3(d0, hb[0][i], hb[1][i], hb[2][i])));
                if (apply_sigmoid) t = up_sigmoid(t);
                v[i] = t;
            }
            store(y, v);
        }
        return;
    }
#pragma unroll 2
    for (int r = 0; r < UP_ROWS; ++r) {
        const int y = ybase + r;
        if (y >= Hout) break;
        int y0, y1; float ly;
        ysrc(y, y0, y1, ly);
        const float hy = 1.f - ly;
        const float* r0 = src + y0 * g;
        const float* r1 = src + y1 * g;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float hx = 1.f - lx[i];
            float t = bilerp(hy, ly, hx, lx[i], r0[x0[i]], r0[x1[i]], r1[x0[i]], r1[x1[i]]);
            if (apply_sigmoid) t = up_sigmoid(t);
            v[i] = t;
        }
        store(y, v);
    }
}

// ---- fused grounding point: argmax over the bilinear-upsampled map WITHOUT writing the map ----
// exp/cxr_pt/inference/grounding_utils.py:166-261 (BlipImageProcessor branch :185-191, flat max + unravel_index :254-259).
// key = (order-preserving float bits << 32) | ~flat_index: atomicMax picks the largest value, lowest index on ties
// (torch.max over the flattened map returns the first maximum).
__device__ __forceinline__ unsigned long long pack_key(float v, unsigned idx) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

__global__ __launch_bounds__(256) void grounding_argmax_kernel(const float* __restrict__ maps, int64_t map_stride,
                                                               unsigned long long* __restrict__ keys, int g, int Hout, int Wout,
                                                               float sy, float sx, int off_y, int off_x) {
    __shared__ unsigned long long red[4];
    const int m = blockIdx.y;
    const float* src = maps + (int64_t)m * map_stride;
    const unsigned total = (unsigned)Hout * (unsigned)Wout;
    unsigned long long best = 0ull;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int y = idx / Wout, x = idx - y * Wout;
        const float fy = src_coord(sy, y, off_y), fx = src_coord(sx, x, off_x);
        int y0 = (int)fy, x0 = (int)fx;
        y0 = y0 > g - 1 ? g - 1 : y0; x0 = x0 > g - 1 ? g - 1 : x0;
        const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float v = bilerp(hy, ly, hx, lx, src[y0 * g + x0], src[y0 * g + x1], src[y1 * g + x0], src[y1 * g + x1]);
        const unsigned long long k = pack_key(v, idx);
        best = k > best ? k : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = red[0];
        for (int i = 1; i < 4; ++i) b = red[i] > b ? red[i] : b;
        atomicMax(keys + m, b);
    }
}

__global__ void grounding_decode_kernel(const unsigned long long* __restrict__ keys, int* __restrict__ xy, int n, int Wout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull);
    xy[2 * i] = (int)(idx % (unsigned)Wout);       // x = w_index
    xy[2 * i + 1] = (int)(idx / (unsigned)Wout);   // y = h_index
}

hipError_t launch_grounding_points(const float* maps, int64_t map_stride, unsigned long long* keys_ws, int* xy_out, int M, int g,
                                   int Hout, int Wout, int keep_aspect, hipStream_t s) {
    if (M <= 0 || g <= 0 || Hout <= 0 || Wout <= 0 || (int64_t)Hout * Wout > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(keys_ws, 0, (size_t)M * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    // keep_aspect: the map is upsampled to the padded square max(H, W) and the original area cropped out
    // (AspectRatioBlipImageProcessor branch, grounding_utils.py:172-190 / segmentation_utils.py:41-60)
    const int P = Hout > Wout ? Hout : Wout;
    const float sy = keep_aspect ? (float)g / (float)P : (float)g / (float)Hout;
    const float sx = keep_aspect ? (float)g / (float)P : (float)g / (float)Wout;
    const int off_y = keep_aspect ? (P - Hout) / 2 : 0, off_x = keep_aspect ? (P - Wout) / 2 : 0;
    const int64_t total = (int64_t)Hout * Wout;
    int nblk = (int)((total + 256 * 16 - 1) / (256 * 16));
    nblk = nblk < 1 ? 1 : (nblk > 256 ? 256 : nblk);
    hipLaunchKernelGGL(grounding_argmax_kernel, dim3(nblk, M), dim3(256), 0, s, maps, map_stride, keys_ws, g, Hout, Wout, sy, sx, off_y, off_x);
    hipLaunchKernelGGL(grounding_decode_kernel, dim3((M + 255) / 256), dim3(256), 0, s, keys_ws, xy_out, M, Wout);
    return hipGetLastError();
}

hipError_t launch_upsample_bilinear(const float* maps, int64_t map_stride, float* out, int64_t* argmax_out, int M, int g,
                                    int Hout, int Wout, int apply_sigmoid, int keep_aspect, hipStream_t s) {
    if (M <= 0 || g <= 0 || Hout <= 0 || Wout <= 0 || argmax_out != nullptr) return hipErrorInvalidValue;
    const int P = Hout > Wout ? Hout : Wout;
    const float sy = keep_aspect ? (float)g / (float)P : (float)g / (float)Hout;
    const float sx = keep_aspect ? (float)g / (float)P : (float)g / (float)Wout;
    const int off_y = keep_aspect ? (P - Hout) / 2 : 0, off_x = keep_aspect ? (P - Wout) / 2 : 0;
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3((Wout + 1023) / 1024, (Hout + UP_ROWS - 1) / UP_ROWS, M), dim3(256), 0, s, maps, map_stride, out, g,
                       Hout, Wout, sy, sx, apply_sigmoid, off_y, off_x);
    return hipGetLastError();
}

}  // namespace rz
