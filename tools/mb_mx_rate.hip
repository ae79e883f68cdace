// Probe (GPU box): sustained rate of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3) against v_mfma_f32_16x16x32_f16 on the whole chip, random
// operands in registers, 8 independent accumulators per wave, 1 or 2 waves per SIMD — what the chip's clock / power management leaves of the
// instruction's 2x per-clock rate.   hipcc --offload-arch=gfx950 -O2 tools/mb_mx_rate.hip -o /tmp/mx_rate && /tmp/mx_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void rate(const int* src, float* out, int iters, unsigned long long* clk) {
    const int l = threadIdx.x + blockIdx.x * blockDim.x;
    v8i a[2], b[4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) a[i][j] = src[(l * 61 + i * 8 + j) & 0xffff];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = src[(l * 67 + 100 + i * 8 + j) & 0xffff];
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
    int sc = 0x7f7f7f7f;
    asm volatile("" : "+v"(sc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (KIND == 0) {
                    const h8 x0 = __builtin_bit_cast(h8, __builtin_shufflevector(a[i], a[i], 0, 1, 2, 3)), x1 = __builtin_bit_cast(h8, __builtin_shufflevector(a[i], a[i], 4, 5, 6, 7));
                    const h8 y0 = __builtin_bit_cast(h8, __builtin_shufflevector(b[j], b[j], 0, 1, 2, 3)), y1 = __builtin_bit_cast(h8, __builtin_shufflevector(b[j], b[j], 4, 5, 6, 7));
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x0, y0, acc[i * 4 + j], 0, 0, 0);
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1, y1, acc[i * 4 + j], 0, 0, 0);
                } else {
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i * 4 + j], 0, 0, 0, sc, 0, sc);
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[l] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    std::vector<int> h(65536);
    srand(1);
    for (auto& v : h) {      // random e4m3-safe bytes / f16-safe halves: exponent fields away from NaN / inf
        unsigned x = 0;
        for (int b = 0; b < 4; ++b) x |= (unsigned)((rand() & 0x80) | (0x20 + (rand() % 0x30))) << (8 * b);
        v = (int)x;
    }
    int* d; float* o; unsigned long long* c;
    (void)hipMalloc(&d, 65536 * 4); (void)hipMalloc(&o, 256 * 8 * 256 * 4); (void)hipMalloc(&c, 4096 * 16);
    (void)hipMemcpy(d, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wgs : {256, 512}) {
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                else hipLaunchKernelGGL(rate<1>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> hc(2 * wgs);
                (void)hipMemcpy(hc.data(), c, 16 * wgs, hipMemcpyDeviceToHost);
                const double cyc = (double)hc[0], us = (double)hc[1] / 100.0;
                // f16-equivalent MACs: kind 0: 16 MFMAs x 16x16x32 per iteration; kind 1: 8 x 16x16x128 (= 32 f16 MFMAs' worth of K)
                const double macs = (double)wgs * 4 * iters * (kind == 0 ? 16.0 * 8192 : 8.0 * 32768);
                if (rep == 2) printf("%3d WGs x 4 waves  %s: %8.3f ms  %7.1f TMAC*2/s  clock %.2f GHz  cycles per MFMA %.1f\n", wgs, kind ? "MX e4m3 16x16x128" : "f16 16x16x32     ",
                                     ms, 2 * macs / ms / 1e9, cyc / us / 1e3, cyc / iters / (kind == 0 ? 16 : 8) / (wgs == 512 ? 2 : 1));
            }
        }
    }
    return 0;
}
