// Micro-benchmark (GPU box): how do the waves of ONE SIMD share its matrix pipe and its vector issue?  The attention loops here are, per wave and
// 64-key tile, a block of MFMAs (bf16 kernel: 36 x 16x16x32) followed by a block of dependent vector work (32 v_exp_f32 + 16 v_cvt_pk_bf16_f32 + a few),
// and the kernels run at ~60 % of either bound.  Modes (one workgroup per CU, W waves per SIMD):
//   0  every wave: [MFMA block ; VALU block] free-running, no barriers               (what the shipped kernels do; W = 1, 2, 3)
//   1  MFMA blocks only            2  VALU blocks only                                  (the two bounds)
//   3  two waves per SIMD in OPPOSITE phases, enforced by one s_barrier per phase: waves 0-3 run MFMA while waves 4-7 run VALU, then swap ("ping-pong")
//   4  two waves per SIMD in the SAME phase with the same barriers                      (the convoy, for comparison)
//   hipcc --offload-arch=gfx950 -O3 tools/mb_mfma_valu.hip -o /tmp/mb_mv && /tmp/mb_mv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NM>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[8], const bf16x8& a, const bf16x8& b) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
}
// 32 exponentials of values derived from the accumulators + 16 packed converts, results folded back so nothing is dead
__device__ __forceinline__ void valu_block(f32x4 (&s)[8], unsigned (&p)[16]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(s[i][r]);
        const bf16x2 lo = __builtin_convertvector((f32x2){e[0], e[1]}, bf16x2), hi = __builtin_convertvector((f32x2){e[2], e[3]}, bf16x2);
        p[2 * i] ^= __builtin_bit_cast(unsigned, lo);
        p[2 * i + 1] ^= __builtin_bit_cast(unsigned, hi);
    }
}

template <int MODE>
__global__ __launch_bounds__(768) void k(float* sink, int iters, float seed) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[8], s[8];
    unsigned p[16];
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (lane + i)); b[i] = (__bf16)(seed * (lane - i)); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = (f32x4){0, 0, 0, 0}; s[i] = (f32x4){-1.f - lane * seed, -2.f, -3.f * seed, -0.5f}; }
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = 0;
    const bool second = wave >= 4;           // waves 4-7 (8-wave launches): the partner wave of each SIMD
    const unsigned long long c0 = __builtin_readcyclecounter();      // s_memtime: shader cycles
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { mfma_block<36>(acc, a, b); asm volatile("" ::: "memory"); valu_block(s, p); asm volatile("" ::: "memory"); }
        if (MODE == 1) { mfma_block<36>(acc, a, b); asm volatile("" ::: "memory"); }
        if (MODE == 2) { valu_block(s, p); asm volatile("" ::: "memory"); }
        if (MODE == 3) {
            if (!second) mfma_block<36>(acc, a, b); else valu_block(s, p);
            __builtin_amdgcn_s_barrier();
            if (second) mfma_block<36>(acc, a, b); else valu_block(s, p);
            __builtin_amdgcn_s_barrier();
        }
        if (MODE == 4) {
            mfma_block<36>(acc, a, b);
            __builtin_amdgcn_s_barrier();
            valu_block(s, p);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(s[i][0]), "+v"(s[i][1]), "+v"(s[i][2]), "+v"(s[i][3])); }        // keeps the 32 exponentials loop-variant (no hoisting), no extra instructions
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (blockIdx.x == 7 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(sink)[64] = c1 - c0;
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += (float)p[i];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if (lds[threadIdx.x] == 77 && seed == 0.123f) sink[0] = 1.f;     // keeps the LDS allocation (one workgroup per CU)
}

template <int MODE>
static void run(const char* what, int waves_per_simd, int iters) {
    float* sink; (void)hipMalloc(&sink, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 120 * 1024, 0, sink, iters, 0.001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            // one "tile" = one MFMA block + one VALU block of ONE wave (modes 1 / 2: one block); tiles per SIMD = waves_per_simd * iters (mode 3 / 4: 1 per wave per iteration)
            const double tiles = (double)waves_per_simd * iters;
            unsigned long long cyc = 0;
            (void)hipMemcpy(&cyc, reinterpret_cast<char*>(sink) + 512, 8, hipMemcpyDeviceToHost);
            printf("%-58s %d wave(s)/SIMD: %8.3f ms  %7.1f ns and %6.0f shader cycles per wave-tile per SIMD (clock %.2f GHz)\n", what, waves_per_simd, ms, ms * 1e6 / tiles,
                   (double)cyc / tiles, (double)cyc / (ms * 1e6));
        }
    }
    (void)hipFree(sink);
}

int main() {
    const int iters = 20000;
    run<1>("MFMA blocks only (36 x 16x16x32 = 576 pipe cycles)", 1, iters);
    run<1>("MFMA blocks only", 2, iters);
    run<2>("VALU blocks only (32 exp + 16 cvt_pk + 16 xor)", 1, iters);
    run<2>("VALU blocks only", 2, iters);
    run<0>("[MFMA ; VALU] per wave, free-running", 1, iters);
    run<0>("[MFMA ; VALU] per wave, free-running", 2, iters);
    run<0>("[MFMA ; VALU] per wave, free-running", 3, iters);
    run<3>("ping-pong: opposite phases, s_barrier per phase", 2, iters);
    run<4>("same phase, s_barrier per phase (convoy)", 2, iters);
    return 0;
}
