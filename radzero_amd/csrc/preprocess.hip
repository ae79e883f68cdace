// radzero_hip — device-side image preprocessing, integer/byte work, gfx950.
// Replaces the CPU DataLoader path exp/cxr_pt/inference/dataset.py:31-51 (cv2 NORM_MINMAX -> 8 bit) followed by the Blip
// image processor (convert RGB, PIL bicubic resize of the uint8 image, rescale 1/255, normalise; processing.py:31-49, :90-91).
//   minmax_kernel   : global min / max of the raw pixels (any of u8 / u16 / f32), all channels together (cv2.normalize)
//   to8_kernel      : v8 = round_half_even(v * scale + shift), saturated to [0, 255]
//   resample_h/v    : Pillow's 8-bit separable resampling (ImagingResample): coefficient tables in 22-bit fixed point are
//                     built on the host exactly as Pillow builds them; horizontal pass first, uint8 intermediate
//   normalize_kernel: out[c][y][x] = (v8 / 255 - mean[c]) / std[c], grey replicated to 3 channels
#include <algorithm>
#include <cstring>

#include "rz_common.h"
#include "rz_kernels.h"

namespace rz {

__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

template <typename S> __device__ __forceinline__ float load_px(const S* p, int64_t i) { return (float)p[i]; }

template <typename S>
__global__ __launch_bounds__(256) void minmax_kernel(const S* __restrict__ img, int64_t n, unsigned* __restrict__ mm) {
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = load_px(img, i);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(mm, f2ord(lo)); atomicMax(mm + 1, f2ord(hi)); }
}

template <typename S>
__global__ __launch_bounds__(256) void to8_kernel(const S* __restrict__ img, int64_t n, const unsigned* __restrict__ mm,
                                                  unsigned char* __restrict__ out) {
#pragma clang fp contract(off)
    const float smin = ord2f(mm[0]), smax = ord2f(mm[1]);
    // cv2.normalize(NORM_MINMAX, alpha=0, beta=255): scale = 255 / (smax - smin) (0 when the image is constant), shift = -smin*scale
    // Exact ties DO occur ((v - min) * 255 / (max - min) = k + 1/2 for integer data), and which way they fall depends on the rounding of the
    // intermediate products: the form pinned here is the documented restatement radzero_amd.synthetic.minmax_to_u8 — v * scale and
    // min * scale each rounded to double, then subtracted (NO fused multiply-add: hipcc contracts by default, and HIP's __dmul_rn / __dsub_rn are
    // plain operators compiled with that default, so they do not help: contraction is switched off for this function by the pragma below).  cv2 itself
    // (float32 multiply-add inside convertTo) is not installed anywhere this code runs: that one step stays unpinned.
    const double scale = (double)(smax - smin) > 2.220446049250313e-16 ? 255.0 / (double)(smax - smin) : 0.0;
    const double lo_scaled = (double)smin * scale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double prod = (double)load_px(img, i) * scale;      // rounded product ...
        const double v = prod - lo_scaled;                           // ... then the subtraction: contraction is off in this function
        int r = (int)rint(v);                          // round half to even (cvRound)
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        out[i] = (unsigned char)r;
    }
}

// horizontal pass: in [rows][in_w][C] u8 -> out [rows][out_w][C] u8
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int rows,
                                                         int in_w, int out_w, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= out_w * C) return;
    const int xx = idx / C, c = idx - xx * C;
    const int y = blockIdx.y;
    const int xmin = bounds[2 * xx], xcnt = bounds[2 * xx + 1];
    const int* k = kk + xx * ksize;
    int ss = 1 << 21;
    const unsigned char* row = in + ((int64_t)y * in_w + xmin) * C + c;
    for (int x = 0; x < xcnt; ++x) ss += (int)row[(int64_t)x * C] * k[x];
    ss >>= 22;
    out[((int64_t)y * out_w + xx) * C + c] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}

// vertical pass: in [in_h][w][C] -> out [out_h][w][C]
__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int w, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= w * C) return;
    const int yy = blockIdx.y;
    const int ymin = bounds[2 * yy], ycnt = bounds[2 * yy + 1];
    const int* k = kk + yy * ksize;
    int ss = 1 << 21;
    const unsigned char* col = in + (int64_t)ymin * w * C + idx;
    for (int y = 0; y < ycnt; ++y) ss += (int)col[(int64_t)y * w * C] * k[y];
    ss >>= 22;
    out[(int64_t)yy * w * C + idx] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}

// u8 [S][S][C] -> fp32 [3][S][S]
__global__ __launch_bounds__(256) void normalize_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int n_px, int C,
                                                        float rescale, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_px) return;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (float)in[(int64_t)i * C + (C == 3 ? c : 0)] * rescale;
        out[(int64_t)c * n_px + i] = (v - mean[c]) / sd[c];
    }
}

template <typename S>
static hipError_t minmax_to8(const S* img, int64_t n, unsigned* mm, unsigned char* out8, hipStream_t s) {
    const unsigned init[2] = {0xFFFFFFFFu, 0u};
    hipError_t e = hipMemcpyAsync(mm, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    const int nb = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(minmax_kernel<S>, dim3(nb), dim3(256), 0, s, img, n, mm);
    hipLaunchKernelGGL(to8_kernel<S>, dim3(nb), dim3(256), 0, s, img, n, mm, out8);
    return hipGetLastError();
}

hipError_t launch_preprocess(const void* img, int src_dtype, int H, int W, int C, int S, const int* bounds_h, const int* kk_h, int ksize_h,
                             const int* bounds_v, const int* kk_v, int ksize_v, const float* mean, const float* stdv, float rescale,
                             unsigned char* ws8, unsigned* mm, float* out, int minmax_normalize, hipStream_t s) {
    if (!img || !out || !ws8 || H <= 0 || W <= 0 || (C != 1 && C != 3) || S <= 0) return hipErrorInvalidValue;
    const int64_t n = (int64_t)H * W * C;
    unsigned char* a8 = ws8;                       // [H][W][C]
    unsigned char* b8 = a8 + n;                    // [H][S][C]
    unsigned char* c8 = b8 + (int64_t)H * S * C;   // [S][S][C]
    hipError_t e = hipSuccess;
    if (minmax_normalize) {
        if (src_dtype == 0) e = minmax_to8((const unsigned char*)img, n, mm, a8, s);
        else if (src_dtype == 1) e = minmax_to8((const unsigned short*)img, n, mm, a8, s);
        else if (src_dtype == 2) e = minmax_to8((const float*)img, n, mm, a8, s);
        else return hipErrorInvalidValue;
        if (e != hipSuccess) return e;
    } else {
        if (src_dtype != 0) return hipErrorInvalidValue;
        e = hipMemcpyAsync(a8, img, n, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(resample_h_kernel, dim3((S * C + 255) / 256, H), dim3(256), 0, s, a8, b8, bounds_h, kk_h, ksize_h, H, W, S, C);
    hipLaunchKernelGGL(resample_v_kernel, dim3((S * C + 255) / 256, S), dim3(256), 0, s, b8, c8, bounds_v, kk_v, ksize_v, S, C);
    hipLaunchKernelGGL(normalize_kernel, dim3((S * S + 255) / 256), dim3(256), 0, s, c8, out, S * S, C, rescale, mean[0], mean[1], mean[2],
                       stdv[0], stdv[1], stdv[2]);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Batched form (round 3): B images of different sizes / dtypes per launch, grid.y (or .z) = image; optional pad-to-square of
// AspectRatioBlipImageProcessor (processing.py:247-259: ImageOps.expand with fill 0 AFTER the 8-bit conversion and the grey->RGB
// replication, so padding the single grey channel with 0 is the same image).  One descriptor per image in device memory.
// ---------------------------------------------------------------------------------------------------
struct PreDesc {
    const void* img; int dtype, H, W, C;             // raw image
    int pad_left, pad_top, PH, PW;                    // padded size (= H, W without padding)
    const int* bounds_h; const int* kk_h; int ksize_h;   // PW -> S
    const int* bounds_v; const int* kk_v; int ksize_v;   // PH -> S
    int64_t a8, b8, c8;                               // byte offsets in the workspace: [PH][PW][C], [PH][S][C], [S][S][C]
};

__device__ __forceinline__ float load_any(const PreDesc& d, int64_t i) {
    return d.dtype == 0 ? (float)((const unsigned char*)d.img)[i] : d.dtype == 1 ? (float)((const unsigned short*)d.img)[i] : ((const float*)d.img)[i];
}

__global__ __launch_bounds__(256) void minmax_batch_kernel(const PreDesc* __restrict__ descs, unsigned* __restrict__ mm) {
    const PreDesc d = descs[blockIdx.y];
    const int64_t n = (int64_t)d.H * d.W * d.C;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = load_any(d, i);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0 && n > 0) { atomicMin(mm + 2 * blockIdx.y, f2ord(lo)); atomicMax(mm + 2 * blockIdx.y + 1, f2ord(hi)); }
}

__global__ __launch_bounds__(256) void init_minmax_kernel(unsigned* __restrict__ mm, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { mm[2 * i] = 0xFFFFFFFFu; mm[2 * i + 1] = 0u; }
}

// 8-bit conversion (cv2.normalize NORM_MINMAX, or a plain copy of uint8 data) into the zero-padded square
__global__ __launch_bounds__(256) void to8_pad_batch_kernel(const PreDesc* __restrict__ descs, const unsigned* __restrict__ mm,
                                                            unsigned char* __restrict__ ws, int minmax) {
#pragma clang fp contract(off)
    const PreDesc d = descs[blockIdx.y];
    double scale = 1.0, lo_scaled = 0.0;
    if (minmax) {
        const float smin = ord2f(mm[2 * blockIdx.y]), smax = ord2f(mm[2 * blockIdx.y + 1]);
        scale = (double)(smax - smin) > 2.220446049250313e-16 ? 255.0 / (double)(smax - smin) : 0.0;
        lo_scaled = (double)smin * scale;
    }
    unsigned char* out = ws + d.a8;
    const int64_t n = (int64_t)d.PH * d.PW * d.C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % d.C);
        const int64_t p = i / d.C;
        const int x = (int)(p % d.PW) - d.pad_left, y = (int)(p / d.PW) - d.pad_top;
        int r = 0;
        if (x >= 0 && x < d.W && y >= 0 && y < d.H) {
            const double prod = (double)load_any(d, ((int64_t)y * d.W + x) * d.C + c) * scale;      // see to8_kernel
            const double v = prod - lo_scaled;
            r = (int)rint(v);                          // round half to even (cvRound)
            r = r < 0 ? 0 : (r > 255 ? 255 : r);
        }
        out[i] = (unsigned char)r;
    }
}

__global__ __launch_bounds__(256) void resample_h_batch_kernel(const PreDesc* __restrict__ descs, unsigned char* __restrict__ ws, int S) {
    const PreDesc d = descs[blockIdx.z];
    const int y = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
    if (y >= d.PH || idx >= S * d.C) return;
    const int xx = idx / d.C, c = idx - xx * d.C;
    const int xmin = d.bounds_h[2 * xx], xcnt = d.bounds_h[2 * xx + 1];
    const int* k = d.kk_h + xx * d.ksize_h;
    int ss = 1 << 21;
    const unsigned char* row = ws + d.a8 + ((int64_t)y * d.PW + xmin) * d.C + c;
    for (int x = 0; x < xcnt; ++x) ss += (int)row[(int64_t)x * d.C] * k[x];
    ss >>= 22;
    ws[d.b8 + ((int64_t)y * S + xx) * d.C + c] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}

__global__ __launch_bounds__(256) void resample_v_batch_kernel(const PreDesc* __restrict__ descs, unsigned char* __restrict__ ws, int S) {
    const PreDesc d = descs[blockIdx.z];
    const int yy = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * d.C) return;
    const int ymin = d.bounds_v[2 * yy], ycnt = d.bounds_v[2 * yy + 1];
    const int* k = d.kk_v + yy * d.ksize_v;
    int ss = 1 << 21;
    const unsigned char* col = ws + d.b8 + (int64_t)ymin * S * d.C + idx;
    for (int y = 0; y < ycnt; ++y) ss += (int)col[(int64_t)y * S * d.C] * k[y];
    ss >>= 22;
    ws[d.c8 + (int64_t)yy * S * d.C + idx] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}

__global__ __launch_bounds__(256) void normalize_batch_kernel(const PreDesc* __restrict__ descs, const unsigned char* __restrict__ ws,
                                                              float* __restrict__ out, int n_px, float rescale, float m0, float m1, float m2,
                                                              float s0, float s1, float s2) {
    const PreDesc d = descs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_px) return;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    const unsigned char* in = ws + d.c8;
    float* o = out + (int64_t)blockIdx.y * 3 * n_px;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (float)in[(int64_t)i * d.C + (d.C == 3 ? c : 0)] * rescale;
        o[(int64_t)c * n_px + i] = (v - mean[c]) / sd[c];
    }
}

// The descriptors travel to the device as KERNEL ARGUMENTS, 16 at a time (1792 bytes per launch): the runtime copies kernel arguments
// when the launch is enqueued, so the caller's array may be freed as soon as the call returns, the host never waits for the stream (an
// hipMemcpyAsync from pageable memory does: it is staged synchronously, i.e. only once the stream has reached it), and the launches
// can be captured into a hipGraph.
struct PreDescChunk { PreDesc d[16]; };
__global__ void put_descs_kernel(PreDescChunk c, PreDesc* __restrict__ dst, int first, int n) {
    const int i = threadIdx.x;
    if (i < 16 && first + i < n) dst[first + i] = c.d[i];
}

size_t preprocess_batch_desc_bytes(int n) { return ((size_t)n * sizeof(PreDesc) + 8 * (size_t)n + 255) / 256 * 256; }

// descs_host: n descriptors with a8 / b8 / c8 already laid out BEHIND the descriptor block (preprocess_batch_desc_bytes(n)); ws: the
// workspace (descriptor block first: [n PreDesc][n x 2 u32 min/max]).
hipError_t launch_preprocess_batch(const void* descs_host, int n, int max_ph, int S, const float* mean, const float* stdv, float rescale,
                                   unsigned char* ws, float* out, int minmax_normalize, hipStream_t s) {
    if (!descs_host || !ws || !out || n <= 0 || S <= 0 || max_ph <= 0) return hipErrorInvalidValue;
    for (int first = 0; first < n; first += 16) {
        PreDescChunk c;
        memset(&c, 0, sizeof c);
        memcpy(c.d, reinterpret_cast<const PreDesc*>(descs_host) + first, (size_t)std::min(16, n - first) * sizeof(PreDesc));
        hipLaunchKernelGGL(put_descs_kernel, dim3(1), dim3(64), 0, s, c, reinterpret_cast<PreDesc*>(ws), first, n);
    }
    const PreDesc* dd = reinterpret_cast<const PreDesc*>(ws);
    unsigned* mm = reinterpret_cast<unsigned*>(ws + (size_t)n * sizeof(PreDesc));
    if (minmax_normalize) {
        hipLaunchKernelGGL(init_minmax_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mm, n);
        hipLaunchKernelGGL(minmax_batch_kernel, dim3(256, n), dim3(256), 0, s, dd, mm);
    }
    hipLaunchKernelGGL(to8_pad_batch_kernel, dim3(256, n), dim3(256), 0, s, dd, mm, ws, minmax_normalize);
    hipLaunchKernelGGL(resample_h_batch_kernel, dim3((S * 3 + 255) / 256, max_ph, n), dim3(256), 0, s, dd, ws, S);
    hipLaunchKernelGGL(resample_v_batch_kernel, dim3((S * 3 + 255) / 256, S, n), dim3(256), 0, s, dd, ws, S);
    hipLaunchKernelGGL(normalize_batch_kernel, dim3((S * S + 255) / 256, n), dim3(256), 0, s, dd, ws, out, S * S, rescale, mean[0], mean[1],
                       mean[2], stdv[0], stdv[1], stdv[2]);
    return hipGetLastError();
}

}  // namespace rz
