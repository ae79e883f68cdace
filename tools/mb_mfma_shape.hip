// Micro-benchmark (GPU box): would the bf16 attention loop gain from 32x32x16 MFMAs?  The loop is bound by the SIMD's vector ISSUE, not by the matrix pipe
// (tools/mb_mfma_valu.hip): an MFMA holds the issue port for 8 cycles whatever its shape, so 16 x 32x32x16 (+ 4 x 16x16x32 for the row sums, fed with a 0/1
// selector fragment) issue for 160 cycles where 36 x 16x16x32 issue for 288, at the same 576 pipe cycles.  Against that: MI355X_MICROARCH.md measures bare
// 32x32x16 loops at a LOWER clock than 16x16x32 loops (power).  This benchmark runs both instruction mixes with the attention loop's vector block (32 v_exp_f32 +
// 16 v_cvt_pk_bf16_f32) on random operand fragments (consecutive MFMAs see different bits: realistic toggling), 1 / 2 / 3 waves per SIMD,
// one workgroup per CU, and reports wall time per wave-tile.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_mfma_shape.hip -o /tmp/mb_ms && /tmp/mb_ms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void valu_block(const float (&s)[32], unsigned (&p)[16]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(s[4 * i + r]);
        const bf16x2 lo = __builtin_convertvector((f32x2){e[0], e[1]}, bf16x2), hi = __builtin_convertvector((f32x2){e[2], e[3]}, bf16x2);
        p[2 * i] = __builtin_bit_cast(unsigned, lo);
        p[2 * i + 1] = __builtin_bit_cast(unsigned, hi);
    }
}

// SHAPE 0: 36 x 16x16x32 (16 scores, 16 P V, 4 row sums)     SHAPE 1: 8 + 8 x 32x32x16 and 4 x 16x16x32     VALU: with the vector block or without
template <int SHAPE, bool VALU>
__global__ __launch_bounds__(768) void k(const bf16x8* __restrict__ frags, float* sink, int iters) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    bf16x8 f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = frags[(i * 64 + lane + 7 * blockIdx.x) & 4095];
    float s[32];
    unsigned p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = 0;
    f32x4 o4[8], l4[2];
    f32x16 o16[2], s16[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) o4[i] = (f32x4){0, 0, 0, 0};
    l4[0] = l4[1] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) { o16[i][j] = 0.f; s16[i][j] = 0.f; }
    // C operand of every first score MFMA (as the kernels do it: no per-tile v_mov); -40: the 2^S stay tiny, finite and with random mantissas
    f32x4 c4 = (f32x4){-40.f, -40.f, -40.f, -40.f};
    f32x16 c16;
#pragma unroll
    for (int j = 0; j < 16; ++j) c16[j] = -40.f;
    asm volatile("" : "+v"(c4), "+v"(c16));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(f[i]));          // opaque: nothing of the tile is loop invariant, no instruction emitted
        if constexpr (SHAPE == 0) {
            f32x4 sa[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) sa[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i & 3], f[4 + (i >> 2)], c4, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) sa[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[4 + (i & 3)], f[i >> 2], sa[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 32; ++i) s[i] = sa[i >> 2][i & 3];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) s16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], f[i + 3], c16, 0, 0, 0);
#pragma unroll
            for (int i = 2; i < 8; ++i) s16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i & 7], f[(i + 3) & 7], s16[i & 1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 32; ++i) s[i] = s16[i >> 4][i & 15];
        }
        bf16x8 pf[4];
        if constexpr (VALU) {
            valu_block(s, p);
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[i] = __builtin_bit_cast(bf16x8, (uint4){p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]});
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[i] = __builtin_bit_cast(bf16x8, (float4){s[8 * i], s[8 * i + 1], s[8 * i + 2], s[8 * i + 3]});
        }
        if constexpr (SHAPE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) o4[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i & 7], pf[i >> 2], o4[i & 7], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) o16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i & 7], pf[i >> 1], o16[i & 1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) l4[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], pf[i], l4[i & 1], 0, 0, 0);
    }
    float r = l4[0][0] + l4[1][1];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += o4[i][0] + o4[i][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) r += o16[i][0] + o16[i][15] + s16[i][3];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if (lds[threadIdx.x] == 77 && iters == -1) sink[0] = 1.f;     // keeps the LDS allocation (one workgroup per CU)
}

template <int SHAPE, bool VALU>
static double run(const char* what, int waves_per_simd, int iters, const bf16x8* frags) {
    float* sink; (void)hipMalloc(&sink, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    (void)hipFuncSetAttribute((const void*)k<SHAPE, VALU>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    double ns = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, VALU>), dim3(256), dim3(threads), 120 * 1024, 0, frags, sink, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        ns = ms * 1e6 / ((double)waves_per_simd * iters);
    }
    printf("%-78s %d wave(s)/SIMD: %7.1f ns per wave-tile per SIMD\n", what, waves_per_simd, ns);
    (void)hipFree(sink);
    return ns;
}

int main() {
    std::vector<unsigned short> h(4096 * 8);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x807f) | 0x3f00); }    // random sign + mantissa, |value| in [0.5, 1)
    bf16x8* frags; (void)hipMalloc(&frags, h.size() * 2);
    (void)hipMemcpy(frags, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int w = 1; w <= 3; ++w) {
        run<0, false>("36 x 16x16x32, no vector block (matrix pipe: 576 cycles)", w, iters, frags);
        run<1, false>("16 x 32x32x16 + 4 x 16x16x32, no vector block (576 cycles)", w, iters, frags);
        const double a = run<0, true>("36 x 16x16x32 + 32 v_exp + 16 v_cvt_pk (the shipped loop's arithmetic)", w, iters, frags);
        const double b = run<1, true>("16 x 32x32x16 + 4 x 16x16x32 + 32 v_exp + 16 v_cvt_pk", w, iters, frags);
        printf("    -> 32x32x16 mix / 16x16x32 mix = %.3f\n", b / a);
    }
    return 0;
}
