// Micro-benchmark (measurement tool): do LDS-DMA staging (global_load_lds) and MFMA work fed by ds_read_b128 overlap on
// one CU, or do they serialise?  One 512-thread workgroup per CU, gemm_kernel_v3-like step: 64 global_load_lds of 1 KB
// (64 KB stage) per step + per wave 24 ds_read_b128 and 64 MFMA 16x16x32.  Modes: 1 = staging only, 2 = compute only,
// 3 = both (2-stage: loads of step k+1 issued before compute k, vmcnt(0)+barrier after).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const char* A, int64_t lda, int nk, int iters, float* sink) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    auto stage = [&](const char* ga, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int piece = wave * 8 + i;
            const int r = piece * 8 + (lane >> 3);
            const char* src = ga + (int64_t)(r & 255) * lda + (int64_t)kt * 128 + ((lane & 7) << 4) + (r >> 8) * 4096;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + buf * 65536 + piece * 1024), 16, 0, 0);
        }
    };
    for (int it = 0; it < iters; ++it) {
        const char* ga = A + (int64_t)((blockIdx.x + it * gridDim.x) % 160) * 256 * lda;
        if (MODE & 1) { stage(ga, 0, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if ((MODE & 1) && kt + 1 < nk) stage(ga, kt + 1, buf ^ 1);
            if (MODE & 2) {
                const char* sa = lds + buf * 65536 + (wave >> 2) * 16384;
                const char* sb = lds + buf * 65536 + 32768 + (wave & 3) * 8192;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 fa[8], fb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const bf16x8*>(sb + (i * 16 + (lane & 15)) * 128 + ((ks * 4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7)) * 16);
#pragma unroll
                    for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sa + (i * 16 + (lane & 15)) * 128 + ((ks * 4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7)) * 16);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 123.456f) sink[0] = s;
}

template <int MODE> static float run(const char* A, int64_t lda, int nk, int iters, float* sink) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, A, lda, nk, iters, sink);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
    }
    return ms;
}

int main() {
    const int64_t lda = 1536;     // K = 768 bf16
    char* A; float* sink;
    (void)hipMalloc(&A, (size_t)43008 * 3072 * 2); (void)hipMalloc(&sink, 4);
    (void)hipMemset(A, 0x3c, (size_t)43008 * 3072 * 2);     // bf16 0x3c3c ~ 0.0115: non-zero operands
    const int nk = 12, iters = 8;
    const float s = run<1>(A, lda, nk, iters, sink), c = run<2>(A, lda, nk, iters, sink), b = run<3>(A, lda, nk, iters, sink);
    const double flops = 256.0 * iters * nk * 2.0 * 256 * 256 * 64;
    printf("staging only : %.3f ms (%.2f us/step, %.1f GB/s/CU)\n", s, s * 1e3 / (iters * nk), 65536.0 * iters * nk / s / 1e6);
    printf("compute only : %.3f ms (%.2f us/step, %.0f TFLOP/s)\n", c, c * 1e3 / (iters * nk), flops / c / 1e9);
    printf("both         : %.3f ms (%.2f us/step, %.0f TFLOP/s)  sum of parts %.3f ms, max of parts %.3f ms\n", b, b * 1e3 / (iters * nk),
           flops / b / 1e9, s + c, s > c ? s : c);
    return 0;
}
