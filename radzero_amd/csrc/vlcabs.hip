// radzero_hip — VL-CABS head (the reference's own arithmetic: exp/cxr_pt/model/losses.py:71-105 and
// SimilarityLogit :187-240, final scaling exp/cxr_pt/model/modeling.py:311-328), always in fp32.
//
//   vhat  = l2norm(LN_shared(tokens))                     (losses.py:90-91, :213)      [scores kernel]
//   S     = qhat . vhat / tau                             (losses.py:219-221)          [scores kernel]
//   p     = softmax_N(S); agg = p . vhat                  (losses.py:222-224)          [partial + finalize]
//   logit = qhat . agg/||agg||                            (losses.py:226-233)          [finalize]
//
// Softmax over N tokens is split into 128-token chunks (online-softmax partials m, l, agg[D]) that the
// finalize kernel merges, so the token tensor is streamed once per stage and nothing of size N x N exists.
// Also: bilinear similarity-map upsample (exp/cxr_pt/inference/segmentation_utils.py:62-70).
#include "rz_common.h"
#include "rz_kernels.h"

namespace rz {

constexpr int VC_ROWS = 64;     // token rows per workgroup in the scores kernel
constexpr int VC_CHUNK = 128;   // token rows per softmax chunk
constexpr int VC_TG = 16;       // prompts per workgroup in the partial kernel

// ---- scores ----
// A workgroup handles VC_ROWS token rows in sub-blocks of 16.  Per sub-block: each wave normalises four rows in registers
// (shared LayerNorm, then L2), writes vhat to global memory and to an LDS tile [16][768] (row stride 772 floats);
// then S[16 tokens][16 prompts] tiles are computed on the matrix pipe with the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32):
// A = vhat rows from LDS, B = qhat rows straight from global/L2 (T x 768 fp32, cached), one 16-byte read of each per
// 16-wide K block feeding four MFMAs.  The former version reduced every (token, prompt) dot product with a 6-step
// cross-lane sum of its own: 0.50 ms at T=14 and 1.7 ms at T=64 against 0.2 ms of HBM time for tokens + vhat.
constexpr int VC_LDS_STRIDE = 772;      // floats; 772 mod 64 = 4 spreads the 16 rows of an A-fragment read over the banks

__global__ __launch_bounds__(256) void vlcabs_scores_kernel(const float* __restrict__ tokens, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ qhat, float inv_tau,
                                                            float* __restrict__ vhat, float* __restrict__ scores, int T,
                                                            int n_valid, int n_pad) {
    __shared__ __attribute__((aligned(16))) float vs[16 * VC_LDS_STRIDE];
    constexpr int D = 768;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int b = blockIdx.y, row0 = blockIdx.x * VC_ROWS;
    const int ttiles = (T + 15) / 16;
    for (int sub = 0; sub < VC_ROWS / 16; ++sub) {
        const int tok0 = row0 + sub * 16;
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int rl = wave * 4 + rr;
            const int64_t row = (int64_t)b * n_pad + tok0 + rl;
            f32x4 v[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const f32x4*>(tokens + row * D + (lane + 64 * i) * 4);
            // shared LayerNorm (eps 1e-5), two-pass
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            const float mu = wave_sum(s) * (1.0f / D);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                v[i] -= mu;
                q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
            }
            const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
            float n2 = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * 4);
                const f32x4 be = *reinterpret_cast<const f32x4*>(beta + (lane + 64 * i) * 4);
                v[i] = v[i] * rstd * g + be;
                n2 += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
            }
            const float inv = 1.0f / fmaxf(sqrtf(wave_sum(n2)), 1e-12f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                v[i] *= inv;
                *reinterpret_cast<f32x4*>(vhat + row * D + (lane + 64 * i) * 4) = v[i];
                *reinterpret_cast<f32x4*>(vs + rl * VC_LDS_STRIDE + (lane + 64 * i) * 4) = v[i];
            }
        }
        __syncthreads();
        // S tile: D[token 4*lg + r][prompt l15]; K is walked in blocks of 16 (lane (., lg) supplies k = 16*kb + 4*lg + u
        // for the u-th of four MFMAs: a permuted but consistent K order for both operands)
        for (int tt = wave; tt < ttiles; tt += 4) {
            const int t = tt * 16 + l15;
            const float* qrow = qhat + (int64_t)(t < T ? t : T - 1) * D + 4 * lg;
            const float* arow = vs + l15 * VC_LDS_STRIDE + 4 * lg;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int kb = 0; kb < D / 16; ++kb) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(arow + kb * 16);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(qrow + kb * 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
            }
            if (t < T) {
                float* o = scores + ((int64_t)b * T + t) * n_valid + tok0 + 4 * lg;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (tok0 + 4 * lg + r < n_valid) o[r] = acc[r] * inv_tau;
            }
        }
        __syncthreads();
    }
}

// ---- per-chunk online-softmax partials: ws[b][chunk][t] = {m, l, agg[768]} ----
__global__ __launch_bounds__(256) void vlcabs_partial_kernel(const float* __restrict__ vhat, const float* __restrict__ scores,
                                                             float* __restrict__ ws, int T, int n_valid, int n_pad) {
    constexpr int D = 768, REC = D + 2;
    __shared__ float p[VC_TG][VC_CHUNK];
    __shared__ float mstat[VC_TG], lstat[VC_TG];
    const int tid = threadIdx.x;
    const int c = blockIdx.x, b = blockIdx.y, t0 = blockIdx.z * VC_TG;
    const int nchunks = gridDim.x;
    const int row0 = c * VC_CHUNK;
    // step 1: 16 threads per prompt, 8 rows each
    {
        const int tl = tid >> 4, sub = tid & 15;
        const int t = t0 + tl;
        float sv[8];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tok = row0 + sub + 16 * i;
            sv[i] = (t < T && tok < n_valid) ? scores[((int64_t)b * T + t) * n_valid + tok] : -INFINITY;
            mx = fmaxf(mx, sv[i]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float e = (mx == -INFINITY) ? 0.f : expf(sv[i] - mx);
            p[tl][sub + 16 * i] = e;
            sum += e;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (sub == 0) { mstat[tl] = mx; lstat[tl] = sum; }
    }
    __syncthreads();
    // step 2: thread owns columns tid, tid+256, tid+512
    float acc[VC_TG][3];
#pragma unroll
    for (int t = 0; t < VC_TG; ++t) acc[t][0] = acc[t][1] = acc[t][2] = 0.f;
    const float* vb = vhat + ((int64_t)b * n_pad + row0) * D;
    for (int r = 0; r < VC_CHUNK; ++r) {
        const float v0 = vb[(int64_t)r * D + tid], v1 = vb[(int64_t)r * D + tid + 256], v2 = vb[(int64_t)r * D + tid + 512];
#pragma unroll
        for (int t = 0; t < VC_TG; ++t) {
            const float pw = p[t][r];
            acc[t][0] = fmaf(pw, v0, acc[t][0]);
            acc[t][1] = fmaf(pw, v1, acc[t][1]);
            acc[t][2] = fmaf(pw, v2, acc[t][2]);
        }
    }
#pragma unroll
    for (int tl = 0; tl < VC_TG; ++tl) {
        const int t = t0 + tl;
        if (t >= T) break;
        float* rec = ws + (((int64_t)b * nchunks + c) * T + t) * REC;
        if (tid == 0) { rec[0] = mstat[tl]; rec[1] = lstat[tl]; }
        rec[2 + tid] = acc[tl][0];
        rec[2 + tid + 256] = acc[tl][1];
        rec[2 + tid + 512] = acc[tl][2];
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
                                                              float* __restrict__ logits, int T, int B, int nchunks) {
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
    if (tid == 0) {
        const float lg = dot / fmaxf(sqrtf(n2), 1e-12f);
        t2i_logits[(int64_t)t * B + b] = lg;
        logits[(int64_t)b * T + t] = lg / tau;
    }
}

size_t vlcabs_workspace_floats(int B, int T, int n_pad, int D) {
    return (size_t)B * (n_pad / VC_CHUNK) * T * (D + 2);
}

hipError_t launch_vlcabs(const float* tokens, const float* ln_gamma, const float* ln_beta, float ln_eps,
                         const float* qhat, float tau, float* vhat, float* ws, float* scores, float* t2i_logits,
                         float* logits, int B, int T, int n_valid, int n_pad, int D, hipStream_t s) {
    if (D != 768 || B <= 0 || T <= 0 || n_pad % VC_CHUNK || n_valid > n_pad) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vlcabs_scores_kernel, dim3(n_pad / VC_ROWS, B), dim3(256), 0, s, tokens, ln_gamma, ln_beta, ln_eps,
                       qhat, 1.0f / tau, vhat, scores, T, n_valid, n_pad);
    const int nchunks = n_pad / VC_CHUNK;
    hipLaunchKernelGGL(vlcabs_partial_kernel, dim3(nchunks, B, (T + VC_TG - 1) / VC_TG), dim3(256), 0, s, vhat, scores, ws, T,
                       n_valid, n_pad);
    hipLaunchKernelGGL(vlcabs_finalize_kernel, dim3(T, B), dim3(256), 0, s, ws, qhat, tau, t2i_logits, logits, T, B, nchunks);
    return hipGetLastError();
}

// ---- bilinear upsample, align_corners=False (F.interpolate semantics), optional sigmoid ----
// One thread produces 4 consecutive pixels of a row (16-byte store); the two source rows of an output row are the
// same for the whole row, so their addresses/weights are computed once per thread.  HBM-write bound:
// n_maps*H*W*4 bytes (4.29 GB at BASELINE cfg 4).
constexpr int UP_ROWS = 8;      // output rows per thread: the four x interpolation set-ups are amortised over them
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
        float fx = sx * (x + off_x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
        int xa = (int)fx;
        xa = xa > g - 1 ? g - 1 : xa;                         // only reachable for x >= Wout (masked below)
        x0[i] = xa;
        x1[i] = xa + (xa < g - 1 ? 1 : 0);
        lx[i] = fx - xa;
    }
    const bool full = x4 + 3 < Wout;
#pragma unroll 2
    for (int r = 0; r < UP_ROWS; ++r) {
        const int y = ybase + r;
        if (y >= Hout) break;
        float fy = sy * (y + off_y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
        const int y0 = (int)fy;
        const int y1 = y0 + (y0 < g - 1 ? 1 : 0);
        const float ly = fy - y0, hy = 1.f - ly;
        const float* r0 = src + y0 * g;
        const float* r1 = src + y1 * g;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float hx = 1.f - lx[i];
            float t = hy * (hx * r0[x0[i]] + lx[i] * r0[x1[i]]) + ly * (hx * r1[x0[i]] + lx[i] * r1[x1[i]]);
            if (apply_sigmoid) t = 1.0f / (1.0f + expf(-t));
            v[i] = t;
        }
        const int64_t base = ((int64_t)m * Hout + y) * Wout + x4;
        float* o = out + base;
        if (full && (base & 3) == 0) {
            *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x4 + i < Wout) o[i] = v[i];
        }
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
        float fy = sy * (y + off_y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
        float fx = sx * (x + off_x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
        int y0 = (int)fy, x0 = (int)fx;
        y0 = y0 > g - 1 ? g - 1 : y0; x0 = x0 > g - 1 ? g - 1 : x0;
        const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float v = hy * (hx * src[y0 * g + x0] + lx * src[y0 * g + x1]) + ly * (hx * src[y1 * g + x0] + lx * src[y1 * g + x1]);
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
