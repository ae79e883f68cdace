// radzero_hip — row-wise (HBM-bound) kernels: LayerNorm, im2col for the patch conv, MPNet embeddings,
// masked mean pooling, LN+L2-normalise.  One 64-lane wave per row, 16-byte accesses, two-pass statistics
// held in registers (exact mean/variance, no E[x^2]-E[x]^2 cancellation).
#include "rz_common.h"
#include "rz_kernels.h"

namespace rz {

// ---- row held by one wave: D = 256*NV floats, lane owns float4 chunks lane + 64*i ----
template <int NV> struct RowRegs { f32x4 v[NV]; };

template <int NV> __device__ __forceinline__ RowRegs<NV> load_row(const float* p, int lane) {
    RowRegs<NV> r;
#pragma unroll
    for (int i = 0; i < NV; ++i) r.v[i] = *reinterpret_cast<const f32x4*>(p + (lane + 64 * i) * 4);
    return r;
}

// in-register LayerNorm of a row (biased variance), returns normalised*gamma+beta in place
template <int NV>
__device__ __forceinline__ void row_layernorm(RowRegs<NV>& r, const float* gamma, const float* beta, float eps, int lane) {
    constexpr float invD = 1.0f / (256.0f * NV);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (r.v[i][0] + r.v[i][1]) + (r.v[i][2] + r.v[i][3]);
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        r.v[i] -= mu;
        q += (r.v[i][0] * r.v[i][0] + r.v[i][1] * r.v[i][1]) + (r.v[i][2] * r.v[i][2] + r.v[i][3] * r.v[i][3]);
    }
    const float rstd = rsqrtf(wave_sum(q) * invD + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + (lane + 64 * i) * 4);
        r.v[i] = r.v[i] * rstd * g + b;
    }
}

// TF:dinov2/modeling_dinov2.py:348,353,441 (norm1/norm2/final layernorm); TF:mpnet/modeling_mpnet.py:198,229.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, T* out_t,
                                                        float* out_f32, int64_t rows, const unsigned* __restrict__ run_if) {
    constexpr int D = 256 * NV;
    if (run_if && *run_if == 0) return;      // predicated launch (fp32 mode's overflow guard): wave-uniform
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    RowRegs<NV> r = load_row<NV>(in + row * D, lane);
    row_layernorm<NV>(r, gamma, beta, eps, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + row * D + c) = r.v[i];
        if (out_t)
            *reinterpret_cast<typename Traits<T>::vec4*>(out_t + row * D + c) =
                pack4<T>(r.v[i][0], r.v[i][1], r.v[i][2], r.v[i][3]);
    }
}

// fp32 mode, hi/lo-split path: the LayerNorm output leaves as the next GEMM's A operand [rows][3D] f16 = [hi | lo | hi] (rz_common.h split4)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_split3_kernel(const float* __restrict__ in, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, f16_t* __restrict__ out3, int64_t rows, unsigned* ovf_flag,
                                                               float* __restrict__ out_f32) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    RowRegs<NV> r = load_row<NV>(in + row * D, lane);
    row_layernorm<NV>(r, gamma, beta, eps, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + row * D + (lane + 64 * i) * 4) = r.v[i];      // the text encoder's post-LN residual (MPNet)
        f16x4 hi, lo;
        split4(r.v[i], hi, lo, ovf_flag);
        f16_t* o = out3 + row * 3 * D + (lane + 64 * i) * 4;
        *reinterpret_cast<f16x4*>(o) = hi;
        *reinterpret_cast<f16x4*>(o + D) = lo;
        *reinterpret_cast<f16x4*>(o + 2 * D) = hi;
    }
}

// the same in the MX form (rz_common.h): row = [hi f16 x D | per 64 columns: lo8 x 64, hi8 x 64] = 4 D bytes
template <int NV>
__global__ __launch_bounds__(256) void layernorm_split_mx_kernel(const float* __restrict__ in, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, char* __restrict__ out, int64_t rows, unsigned* ovf_flag) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    RowRegs<NV> r = load_row<NV>(in + row * D, lane);
    row_layernorm<NV>(r, gamma, beta, eps, lane);
    char* o = out + row * 4 * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f16x4 hi;
        uint32_t lo8, hi8;
        split4_mx(r.v[i], hi, lo8, hi8, MX_A_HI_SCALE, MX_A_LO_SCALE, ovf_flag);
        const int k = (lane + 64 * i) * 4;
        *reinterpret_cast<f16x4*>(o + 2 * k) = hi;
        char* pr = o + mx_pair_off(D, k);
        *reinterpret_cast<uint32_t*>(pr) = lo8;
        *reinterpret_cast<uint32_t*>(pr + 64) = hi8;
    }
}

hipError_t launch_layernorm_split3(const float* in, const float* gamma, const float* beta, float eps, void* out3, int64_t rows, int D, unsigned* ovf_flag, hipStream_t s, int mx, float* out_f32) {
    if (D != 768 || rows <= 0 || (mx && out_f32)) return hipErrorInvalidValue;
    if (mx) hipLaunchKernelGGL((layernorm_split_mx_kernel<3>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, in, gamma, beta, eps, (char*)out3, rows, ovf_flag);
    else hipLaunchKernelGGL((layernorm_split3_kernel<3>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, in, gamma, beta, eps, (f16_t*)out3, rows, ovf_flag, out_f32);
    return hipGetLastError();
}

hipError_t launch_layernorm(int dtype, const float* in, const float* gamma, const float* beta, float eps,
                            void* out_t, float* out_f32, int64_t rows, int D, hipStream_t s, const unsigned* run_if) {
    if (D != 768 || rows <= 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    switch (dtype) {
        case DT_F32: hipLaunchKernelGGL((layernorm_kernel<float, 3>), grid, block, 0, s, in, gamma, beta, eps, (float*)out_t, out_f32, rows, run_if); break;
        case DT_BF16: hipLaunchKernelGGL((layernorm_kernel<bf16_t, 3>), grid, block, 0, s, in, gamma, beta, eps, (bf16_t*)out_t, out_f32, rows, run_if); break;
        case DT_F16: hipLaunchKernelGGL((layernorm_kernel<f16_t, 3>), grid, block, 0, s, in, gamma, beta, eps, (f16_t*)out_t, out_f32, rows, run_if); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- fused-LayerNorm support (gemm8.hip "Fused LayerNorm") ----
// ln_finalize: one thread per row merges the 12 (mean, M2) partials of 64 columns each that EPI_RESID_SCALE_LN wrote
// (equal counts: mean = average of means, M2 = sum M2_p + 64 * sum (mean_p - mean)^2 — Chan et al., exact) into (mean, rstd).
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, float* __restrict__ mu_inout, float* __restrict__ stat, float eps, int64_t rows, int mu_is_zero) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(part + row * 24);
    f32x4 v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = p[i];
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) mean += v[i][0] + v[i][2];
    mean *= (1.0f / 12.0f);
    float m2 = 0.f, dev = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        m2 += v[i][1] + v[i][3];
        const float d0 = v[i][0] - mean, d1 = v[i][2] - mean;
        dev += d0 * d0 + d1 * d1;
    }
    const float var = (m2 + 64.0f * dev) * (1.0f / 768.0f);
    // the producer centred its T copy with the row's previous mean: consumers subtract only the remainder
    // (mu_is_zero: the producer had no earlier mean — EPI_PATCH_LN centres with 0 — and mu_inout holds nothing yet)
    *reinterpret_cast<f32x2*>(stat + row * 2) = (f32x2){mean - (mu_is_zero ? 0.f : mu_inout[row]), rsqrtf(var + eps)};
    mu_inout[row] = mean;
}

hipError_t launch_ln_finalize(const float* part, float* mu_inout, float* stat, float eps, int64_t rows, hipStream_t s, bool mu_is_zero) {
    if (rows <= 0 || !mu_inout) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, part, mu_inout, stat, eps, rows, mu_is_zero ? 1 : 0);
    return hipGetLastError();
}

// ln_prepare: what a fused-LayerNorm consumer needs of a residual-stream row that no GEMM epilogue produced: its copy in T,
// times the consuming LayerNorm's gain, and (mean, rstd).  With gamma != nullptr the row is first LayerNorm'ed (eps_in) into out_f32 — the ViT's final LayerNorm,
// whose OUTPUT is the residual stream of the align blocks — and copy / statistics are those of the normalised row.
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_prepare_kernel(const float* __restrict__ in, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps_in, float* out_f32,
                                                         const float* __restrict__ copy_gain, T* __restrict__ copy_t,
                                                         float* __restrict__ mu_out, float* __restrict__ stat, float eps_stat, int64_t rows) {
    constexpr int D = 256 * NV;
    constexpr float invD = 1.0f / D;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    RowRegs<NV> r = load_row<NV>(in + row * D, lane);
    if (gamma) {
        row_layernorm<NV>(r, gamma, beta, eps_in, lane);
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(out_f32 + row * D + (lane + 64 * i) * 4) = r.v[i];
    }
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sm += (r.v[i][0] + r.v[i][1]) + (r.v[i][2] + r.v[i][3]);
    const float mu = wave_sum(sm) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4 d = r.v[i] - mu;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    const float rstd = rsqrtf(wave_sum(q) * invD + eps_stat);
    if (lane == 0) {            // the copy is centred with the exact mean: nothing left for the consumer to subtract
        *reinterpret_cast<f32x2*>(stat + row * 2) = (f32x2){0.f, rstd};
        mu_out[row] = mu;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4 gv = (r.v[i] - mu) * *reinterpret_cast<const f32x4*>(copy_gain + (lane + 64 * i) * 4);
        *reinterpret_cast<typename Traits<T>::vec4*>(copy_t + row * D + (lane + 64 * i) * 4) = pack4<T>(gv[0], gv[1], gv[2], gv[3]);
    }
}

hipError_t launch_ln_prepare(int dtype, const float* in, const float* gamma, const float* beta, float eps_in, float* out_f32,
                             const float* copy_gain, void* copy_t, float* mu_out, float* stat, float eps_stat, int64_t rows, int D, hipStream_t s) {
    if (D != 768 || rows <= 0 || !copy_t || !copy_gain || !stat || !mu_out || (gamma && (!beta || !out_f32))) return hipErrorInvalidValue;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    switch (dtype) {
        case DT_BF16: hipLaunchKernelGGL((ln_prepare_kernel<bf16_t, 3>), grid, block, 0, s, in, gamma, beta, eps_in, out_f32, copy_gain, (bf16_t*)copy_t, mu_out, stat, eps_stat, rows); break;
        case DT_F16: hipLaunchKernelGGL((ln_prepare_kernel<f16_t, 3>), grid, block, 0, s, in, gamma, beta, eps_in, out_f32, copy_gain, (f16_t*)copy_t, mu_out, stat, eps_stat, rows); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- shared LayerNorm + L2 normalisation (exp/cxr_pt/model/losses.py:90-91,163-164 then :212-213) ----
template <int NV>
__global__ __launch_bounds__(256) void ln_l2norm_kernel(const float* __restrict__ in, int64_t ld_in,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, float* __restrict__ out, int64_t rows, int l2) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    RowRegs<NV> r = load_row<NV>(in + row * ld_in, lane);
    row_layernorm<NV>(r, gamma, beta, eps, lane);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) q += (r.v[i][0] * r.v[i][0] + r.v[i][1] * r.v[i][1]) + (r.v[i][2] * r.v[i][2] + r.v[i][3] * r.v[i][3]);
    const float nrm = l2 ? fmaxf(sqrtf(wave_sum(q)), 1e-12f) : 1.0f;   // F.normalize: x / max(||x||, eps); sim_op "dot": LayerNorm only
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(out + row * D + (lane + 64 * i) * 4) = r.v[i] / nrm;
}

hipError_t launch_ln_l2norm(const float* in, int64_t ld_in, const float* gamma, const float* beta, float eps,
                            float* out, int64_t rows, int D, int l2, hipStream_t s) {
    if (D != 768 || rows <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL((ln_l2norm_kernel<3>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, in, ld_in, gamma, beta, eps, out, rows, l2);
    return hipGetLastError();
}

// ---- im2col for Conv2d(3,768,k=14,s=14): TF:dinov2/modeling_dinov2.py:139-148 ----
// out[b][tok][k], tok = 1 + py*gw + px, k = (c*14 + ky)*14 + kx; zero for tok==0, tok>=1+gh*gw, k>=C*P*P.
// One workgroup per (image, patch row): the 14 pixel rows of each channel are read as whole rows (coalesced, 4 KB each at 1024^2),
// regrouped per patch in LDS, and leave as runs of P*P elements per (token, channel) in 4-element stores (a workgroup per token
// read 42 runs of 56 bytes: 0.28 ms per 32 images against 0.13 for the bytes).  blockIdx.x == gh zero-fills the CLS row and the padding rows.
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ px, T* __restrict__ out, int B, int C,
                                                     int Himg, int Wimg, int P, int gh, int gw, int n_pad, int k_pad, const unsigned* __restrict__ run_if) {
    typedef typename Traits<T>::vec4 v4;
    if (run_if && *run_if == 0) return;      // predicated launch (fp32 mode's overflow guard)
    extern __shared__ __attribute__((aligned(16))) char im2col_lds[];
    T* tile = reinterpret_cast<T*>(im2col_lds);          // [gw][pp4]: one channel of one patch row, P*P elements per patch (+ pad to 4)
    const int b = blockIdx.y, tid = threadIdx.x;
    const int np = gh * gw, pp = P * P, pp4 = (pp + 3) & ~3, kreal = C * pp;
    T* obase = out + (int64_t)b * n_pad * k_pad;
    const v4 zero = pack4<T>(0.f, 0.f, 0.f, 0.f);
    if ((int)blockIdx.x == gh) {                          // CLS row + rows past the last patch
        const int units = k_pad / 4, rows = 1 + (n_pad - 1 - np);
        for (int u = tid; u < rows * units; u += 256) {
            const int r = u / units, tok = r == 0 ? 0 : np + r;
            *reinterpret_cast<v4*>(obase + (int64_t)tok * k_pad + (u - r * units) * 4) = zero;
        }
        return;
    }
    const int py = blockIdx.x;
    const int w = gw * P;
    const bool vec_ok = (pp % 4 == 0) && (k_pad % 4 == 0);
    const float inv_p = 1.0f / P, inv_units = 4.0f / pp;
    for (int c = 0; c < C; ++c) {
        const float* src = px + (((int64_t)b * C + c) * Himg + (int64_t)py * P) * Wimg;
        if (P == 14) {                                    // the model's patch size: 14 rows x 2 column chunks of loads in flight per thread
#pragma unroll 2
            for (int x = tid; x < w; x += 256) {
                const int pxx = (int)((x + 0.5f) * inv_p);            // x / P without the integer division (exact below 2^20)
                T* dst = tile + pxx * pp4 + (x - pxx * 14);
                float v[14];
#pragma unroll
                for (int ky = 0; ky < 14; ++ky) v[ky] = src[(int64_t)ky * Wimg + x];
#pragma unroll
                for (int ky = 0; ky < 14; ++ky) dst[ky * 14] = from_f32<T>(v[ky]);
            }
        } else {
            for (int ky = 0; ky < P; ++ky) {
                const float* row = src + (int64_t)ky * Wimg;
                for (int x = tid; x < w; x += 256) {
                    const int pxx = (int)((x + 0.5f) * inv_p);
                    tile[pxx * pp4 + ky * P + (x - pxx * P)] = from_f32<T>(row[x]);
                }
            }
        }
        __syncthreads();
        if (vec_ok) {
            const int units = pp / 4;
            for (int u = tid; u < gw * units; u += 256) {
                const int pxx = (int)((u + 0.5f) * inv_units), j = u - pxx * units;
                *reinterpret_cast<v4*>(obase + (int64_t)(1 + py * gw + pxx) * k_pad + c * pp + j * 4) =
                    *reinterpret_cast<const v4*>(tile + pxx * pp4 + j * 4);
            }
        } else {
            for (int u = tid; u < gw * pp; u += 256) {
                const int pxx = u / pp, j = u - pxx * pp;
                obase[(int64_t)(1 + py * gw + pxx) * k_pad + c * pp + j] = tile[pxx * pp4 + j];
            }
        }
        __syncthreads();
    }
    for (int u = tid; u < gw * (k_pad - kreal); u += 256) {       // K padding of this row's tokens
        const int pxx = u / (k_pad - kreal), j = u - pxx * (k_pad - kreal);
        obase[(int64_t)(1 + py * gw + pxx) * k_pad + kreal + j] = from_f32<T>(0.f);
    }
}

hipError_t launch_im2col(int dtype, const float* px, void* out, int B, int C, int Himg, int Wimg, int patch, int gh,
                         int gw, int n_pad, int k_pad, hipStream_t s, const unsigned* run_if) {
    if (B <= 0 || gh * patch > Himg || gw * patch > Wimg || 1 + gh * gw > n_pad || C * patch * patch > k_pad || k_pad % 4) return hipErrorInvalidValue;
    const int pp4 = (patch * patch + 3) & ~3;
    const size_t es = dtype == DT_F32 ? 4 : 2;
    const size_t lds = (size_t)gw * pp4 * es;
    if (lds > 150 * 1024) return hipErrorInvalidValue;   // 2000+ px wide images in fp32: not a RadZero shape
    dim3 grid(gh + 1, B), block(256);
    if (lds > 64 * 1024) {                                // fp32 above ~1160 px: beyond the default dynamic-LDS limit
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&im2col_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    switch (dtype) {
        case DT_F32: hipLaunchKernelGGL(im2col_kernel<float>, grid, block, lds, s, px, (float*)out, B, C, Himg, Wimg, patch, gh, gw, n_pad, k_pad, run_if); break;
        case DT_BF16: hipLaunchKernelGGL(im2col_kernel<bf16_t>, grid, block, lds, s, px, (bf16_t*)out, B, C, Himg, Wimg, patch, gh, gw, n_pad, k_pad, run_if); break;
        case DT_F16: hipLaunchKernelGGL(im2col_kernel<f16_t>, grid, block, lds, s, px, (f16_t*)out, B, C, Himg, Wimg, patch, gh, gw, n_pad, k_pad, run_if); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- MPNet embeddings: TF:mpnet/modeling_mpnet.py:58-95, position ids :873-881 ----
// position id = (number of non-pad tokens at or before i) * (ids[i] != pad) + pad
template <typename T, int NV>
__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word_emb,
                                                         const float* __restrict__ pos_emb, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float* __restrict__ h,
                                                         T* __restrict__ xn, int Tn, int L, int vocab, int max_pos, int pad_id, const unsigned* __restrict__ run_if) {
    constexpr int D = 256 * NV;
    if (run_if && *run_if == 0) return;      // predicated launch (fp32 mode's text guard)
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)Tn * L) return;
    const int t = (int)(row / L), i = (int)(row % L);
    int cnt = 0;
    for (int jj = lane; jj <= i; jj += 64) cnt += (ids[(int64_t)t * L + jj] != pad_id) ? 1 : 0;
    cnt = (int)wave_sum((float)cnt);
    int64_t id = ids[row];
    int pos = (id != pad_id) ? cnt + pad_id : pad_id;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);        // host validates; clamp keeps the access in bounds
    pos = pos >= max_pos ? max_pos - 1 : pos;
    RowRegs<NV> r = load_row<NV>(word_emb + id * D, lane);
    const RowRegs<NV> pe = load_row<NV>(pos_emb + (int64_t)pos * D, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) r.v[k] += pe.v[k];
    row_layernorm<NV>(r, gamma, beta, eps, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (lane + 64 * k) * 4;
        *reinterpret_cast<f32x4*>(h + row * D + c) = r.v[k];
        *reinterpret_cast<typename Traits<T>::vec4*>(xn + row * D + c) = pack4<T>(r.v[k][0], r.v[k][1], r.v[k][2], r.v[k][3]);
    }
}

hipError_t launch_text_embed(int dtype, const int64_t* ids, const float* word_emb, const float* pos_emb,
                             const float* gamma, const float* beta, float eps, float* h, void* xn, int T, int L, int D,
                             int vocab, int max_pos, int pad_id, hipStream_t s, const unsigned* run_if) {
    if (D != 768 || T <= 0 || L <= 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)(((int64_t)T * L + 3) / 4)), block(256);
    switch (dtype) {
        case DT_F32: hipLaunchKernelGGL((text_embed_kernel<float, 3>), grid, block, 0, s, ids, word_emb, pos_emb, gamma, beta, eps, h, (float*)xn, T, L, vocab, max_pos, pad_id, run_if); break;
        case DT_BF16: hipLaunchKernelGGL((text_embed_kernel<bf16_t, 3>), grid, block, 0, s, ids, word_emb, pos_emb, gamma, beta, eps, h, (bf16_t*)xn, T, L, vocab, max_pos, pad_id, run_if); break;
        case DT_F16: hipLaunchKernelGGL((text_embed_kernel<f16_t, 3>), grid, block, 0, s, ids, word_emb, pos_emb, gamma, beta, eps, h, (f16_t*)xn, T, L, vocab, max_pos, pad_id, run_if); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- masked mean pool: exp/cxr_pt/model/modeling.py:148-156 ----
__global__ __launch_bounds__(256) void masked_meanpool_kernel(const float* __restrict__ h, const int64_t* __restrict__ mask,
                                                              float* __restrict__ out, int L, int D) {
    const int t = blockIdx.x;
    float msum = 0.f;
    for (int i = 0; i < L; ++i) msum += (float)mask[(int64_t)t * L + i];
    const float den = fmaxf(msum, 1e-9f);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float acc = 0.f;
        for (int i = 0; i < L; ++i) acc += h[((int64_t)t * L + i) * D + d] * (float)mask[(int64_t)t * L + i];
        out[(int64_t)t * D + d] = acc / den;
    }
}

hipError_t launch_masked_meanpool(const float* h, const int64_t* mask, float* out, int T, int L, int D, hipStream_t s) {
    if (T <= 0 || L <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(masked_meanpool_kernel, dim3(T), dim3(256), 0, s, h, mask, out, L, D);
    return hipGetLastError();
}

// ---- the alignment heads beside VL-CABS (exp/cxr_pt/model/modeling.py:330-353, :115-117): small fp32 products on tensors the library already
// holds.  Not a hot path (the released configuration computes "radzero" logits): plain wave-per-row kernels, no matrix pipe.
// rows_dot: out[(m / rpg) * og + (m % rpg) * orow + n * ocol] = sum_k a[m][k] b[n][k] (+ bias[n]), K % 4 == 0, fixed summation order.
__global__ __launch_bounds__(256) void rows_dot_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                       const float* __restrict__ bias, float* __restrict__ out, int M, int N, int K, int rpg,
                                                       int64_t og, int64_t orow, int64_t ocol) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + (int64_t)m * lda);
    float* o = out + (int64_t)(m / rpg) * og + (int64_t)(m % rpg) * orow;
    const int K4 = K >> 2;
    for (int n = 0; n < N; ++n) {
        const f32x4* b4 = reinterpret_cast<const f32x4*>(b + (int64_t)n * ldb);
        float acc = 0.f;
        for (int i = lane; i < K4; i += 64) {
            const f32x4 x = a4[i], y = b4[i];
            acc += (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]);
        }
        acc = wave_sum(acc);
        if (lane == 0) o[(int64_t)n * ocol] = acc + (bias ? bias[n] : 0.f);
    }
}

hipError_t launch_rows_dot(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, float* out, int M, int N, int K,
                           int rows_per_group, int64_t out_group_stride, int64_t out_row_stride, int64_t out_col_stride, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 4 || lda % 4 || ldb % 4 || rows_per_group <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rows_dot_kernel, dim3((M + 3) / 4), dim3(256), 0, s, a, lda, b, ldb, bias, out, M, N, K, rows_per_group, out_group_stride,
                       out_row_stride, out_col_stride);
    return hipGetLastError();
}

// image_features (modeling.py:115-117): l2norm([cls | mean over the patch tokens]) per image; tokens [B][image_stride rows of D], row 0 = cls,
// rows 1..n_tokens-1 = patches.  Pass 1: column means of the patch rows (one workgroup per image and 64-column slab); pass 2: concatenate + normalise.
__global__ __launch_bounds__(256) void patch_mean_kernel(const float* __restrict__ tokens, int64_t image_stride, int n_tokens, int D, float* __restrict__ feat) {
    // 16 row groups x 16 lanes of four columns: a workgroup reads 16 rows x 256 contiguous bytes per step
    __shared__ f32x4 red[16][16];
    const int b = blockIdx.y, c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + c4 * 4;
    const float* base = tokens + (int64_t)b * image_stride * D + c;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 1 + rg; r < n_tokens; r += 16) acc += *reinterpret_cast<const f32x4*>(base + (int64_t)r * D);
    red[rg][c4] = acc;
    __syncthreads();
    if (rg == 0) {
        f32x4 sum = red[0][c4];
#pragma unroll
        for (int i = 1; i < 16; ++i) sum += red[i][c4];
        *reinterpret_cast<f32x4*>(feat + (int64_t)b * 2 * D + D + c) = sum * (1.0f / (float)(n_tokens - 1));
        *reinterpret_cast<f32x4*>(feat + (int64_t)b * 2 * D + c) = *reinterpret_cast<const f32x4*>(base);          // the cls token
    }
}
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ x, int len) {
    __shared__ float red[4];
    float* row = x + (int64_t)blockIdx.x * len;
    float q = 0.f;
    for (int i = threadIdx.x; i < len; i += 256) q += row[i] * row[i];
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float inv = 1.0f / fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), 1e-12f);     // F.normalize: x / max(||x||, eps)
    for (int i = threadIdx.x; i < len; i += 256) row[i] *= inv;
}

hipError_t launch_image_features(const float* tokens, int64_t image_stride, int B, int n_tokens, int D, float* out, hipStream_t s) {
    if (B <= 0 || n_tokens < 2 || D <= 0 || D % 64 || image_stride < n_tokens) return hipErrorInvalidValue;
    hipLaunchKernelGGL(patch_mean_kernel, dim3(D / 64, B), dim3(256), 0, s, tokens, image_stride, n_tokens, D, out);
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(B), dim3(256), 0, s, out, 2 * D);
    return hipGetLastError();
}

// ---- compact copy of the valid tokens: [B][Npad][D] -> [B][N][D] ----
__global__ __launch_bounds__(256) void copy_tokens_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_valid,
                                                          int n_pad, int D4, const unsigned* __restrict__ run_if) {
    if (run_if && *run_if == 0) return;      // predicated launch (fp32 mode's overflow guard)
    const int tok = blockIdx.x, b = blockIdx.y;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src) + ((int64_t)b * n_pad + tok) * D4;
    f32x4* d4 = reinterpret_cast<f32x4*>(dst) + ((int64_t)b * n_valid + tok) * D4;
    for (int i = threadIdx.x; i < D4; i += blockDim.x) d4[i] = s4[i];
}

hipError_t launch_copy_tokens(const float* src, float* dst, int B, int n_valid, int n_pad, int D, hipStream_t s, const unsigned* run_if) {
    if (D % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(copy_tokens_kernel, dim3(n_valid, B), dim3(192), 0, s, src, dst, n_valid, n_pad, D / 4, run_if);
    return hipGetLastError();
}

// ---- fp32 mode's overflow guard words (rz_kernels.h launch_guard_word) ----
__global__ void guard_word_kernel(unsigned* __restrict__ words, int op, int flag_idx, int count_idx) {
    if (op == 0) words[flag_idx] = 0;
    else if (words[flag_idx]) words[count_idx] += 1;
}

hipError_t launch_guard_word(unsigned* words, int op, hipStream_t s, int flag_idx, int count_idx) {
    if (!words || (op != 0 && op != 1) || flag_idx < 0 || flag_idx > 7 || count_idx < 0 || count_idx > 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(guard_word_kernel, dim3(1), dim3(1), 0, s, words, op, flag_idx, count_idx);
    return hipGetLastError();
}

}  // namespace rz
