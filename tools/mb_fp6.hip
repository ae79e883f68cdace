// Probe (GPU box), VERDICT r5 item 3 step (b): v_mfma_scale_f32_16x16x128_f8f6f4 with FP6 (e2m3, cbsz = blgp = 2) operands —
//   1. operand layout (which lane / bit field holds which k; which lane's scale byte covers which 32-k block), found with exact data,
//   2. sustained rate against the e4m3 form and the f16 MFMA, alone and in the GEMM's mix (two f16 MFMAs + one scaled MFMA per 64 K).
//   hipcc --offload-arch=gfx950 -O2 tools/mb_fp6.hip -o /tmp/mb_fp6 && /tmp/mb_fp6
// e2m3 (OCP MX FP6): s e1 e0 m2 m1 m0, bias 1: e = 0 -> m / 8 (subnormal), else (1 + m / 8) 2^(e - 1); 0x08 = 1.0, max 7.5.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void mx6(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 2, 2, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}
// A in e2m3, B in e4m3 (mixed formats in one instruction)
__global__ void mx68(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 2, 0, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}

// one-hot scan: A candidate x = (lane group ga = x / 32, element ja = x % 32) has 1.0 in lane 16 ga (row 0); B likewise (col 0): out[x][y] = D[0][0]
__global__ void scan6(const v8i* acand, const v8i* bcand, float* out) {
    const int l = threadIdx.x;
    int one = 0x7f7f7f7f;
    asm volatile("" : "+v"(one));
    for (int x = 0; x < 128; ++x) {
        const v8i a = acand[x * 64 + l];
        for (int y = 0; y < 128; ++y) {
            const v8i b = bcand[y * 64 + l];
            v4f acc = {0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 2, 2, 0, one, 0, one);
            if (l == 0) out[x * 128 + y] = acc[0];
        }
    }
}

// KIND 0: f16 16x16x32; 1: e4m3 16x16x128; 2: e2m3 16x16x128; 3: GEMM mix e4m3 (2 f16 + 1 scaled per 64 K); 4: GEMM mix e2m3.
// Inline asm with pinned operands: left to itself the compiler shuffles the 6-dword sub-tuples through v_accvgpr moves inside the loop.
typedef int v6i __attribute__((ext_vector_type(6)));
#define MF16(ACC, X, Y) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(X), "v"(Y))
#define MFP8(ACC, X, Y, S) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(ACC) : "v"(X), "v"(Y), "v"(S))
#define MFP6(ACC, X, Y, S) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(ACC) : "v"(X), "v"(Y), "v"(S))
#define MFP68(ACC, X, Y, S) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:2" : "+v"(ACC) : "v"(X), "v"(Y), "v"(S))
#define MFP86(ACC, X, Y, S) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] blgp:2" : "+v"(ACC) : "v"(X), "v"(Y), "v"(S))
template <int KIND>
__global__ __launch_bounds__(256) void rate(const int* src, float* out, int iters, unsigned long long* clk) {
    const int l = threadIdx.x + blockIdx.x * blockDim.x;
    v8i a[2], b[4];
    v6i a6[2], b6[4];
    h8 ah[2][2], bh[4][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) a[i][j] = src[(l * 61 + i * 8 + j) & 0xffff];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = src[(l * 67 + 100 + i * 8 + j) & 0xffff];
    for (int i = 0; i < 2; ++i) { for (int j = 0; j < 6; ++j) a6[i][j] = a[i][j]; ah[i][0] = __builtin_bit_cast(h8, __builtin_shufflevector(a[i], a[i], 0, 1, 2, 3)); ah[i][1] = __builtin_bit_cast(h8, __builtin_shufflevector(a[i], a[i], 4, 5, 6, 7)); }
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 6; ++j) b6[i][j] = b[i][j]; bh[i][0] = __builtin_bit_cast(h8, __builtin_shufflevector(b[i], b[i], 0, 1, 2, 3)); bh[i][1] = __builtin_bit_cast(h8, __builtin_shufflevector(b[i], b[i], 4, 5, 6, 7)); }
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
    int sc = 0x7f7f7f7f;
    asm volatile("" : "+v"(sc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        // independent accumulators back to back: each MFMA kind sweeps the 8 (i, j) pairs before the next kind touches the same accumulator
        if constexpr (KIND == 0 || KIND == 3 || KIND == 4 || KIND == 7) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int p = 0; p < 8; ++p) MF16(acc[p], ah[p >> 2][h], bh[p & 3][h]);
        }
        if constexpr (KIND == 1 || KIND == 3) {
#pragma unroll
            for (int p = 0; p < 8; ++p) MFP8(acc[p], a[p >> 2], b[p & 3], sc);
        }
        if constexpr (KIND == 2 || KIND == 4) {
#pragma unroll
            for (int p = 0; p < 8; ++p) MFP6(acc[p], a6[p >> 2], b6[p & 3], sc);
        }
        if constexpr (KIND == 5 || KIND == 7) {      // A e2m3 x B e4m3
#pragma unroll
            for (int p = 0; p < 8; ++p) MFP68(acc[p], a6[p >> 2], b[p & 3], sc);
        }
        if constexpr (KIND == 6) {                   // A e4m3 x B e2m3
#pragma unroll
            for (int p = 0; p < 8; ++p) MFP86(acc[p], a[p >> 2], b6[p & 3], sc);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[l] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static uint8_t e2m3_of_int(int v) {      // v in -4..4
    static const uint8_t pos[5] = {0x00, 0x08, 0x10, 0x14, 0x18};
    return v < 0 ? (uint8_t)(pos[-v] | 0x20) : pos[v];
}
static uint8_t e4m3_of_int(int v) {
    static const uint8_t pos[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    return v < 0 ? (uint8_t)(pos[-v] | 0x80) : pos[v];
}
// element j (0..31) of a lane's FP6 operand: bits 6 j .. 6 j + 5 of the lane's first 192 bits (little endian over the 6 dwords)
static void put6(uint8_t* lane32, int j, uint8_t code) {
    const int bit = 6 * j;
    for (int t = 0; t < 6; ++t) {
        const int bb = bit + t;
        if ((code >> t) & 1) lane32[bb >> 3] |= (uint8_t)(1u << (bb & 7));
        else lane32[bb >> 3] &= (uint8_t)~(1u << (bb & 7));
    }
}

int main() {
    void *da, *db, *dc, *dsa, *dsb, *dscan, *dcand;
    (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dc, 64 * 16); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256);
    (void)hipMalloc(&dscan, 128 * 128 * 4);
    std::vector<uint8_t> A(2048, 0), B(2048, 0);
    std::vector<uint32_t> SA(64, 0x7f7f7f7f), SB(64, 0x7f7f7f7f);
    std::vector<float> D(256);
    auto run = [&](int mixed) {
        (void)hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
        (void)hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice);
        if (mixed) hipLaunchKernelGGL(mx68, dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (v4f*)dc, (const int*)dsa, (const int*)dsb);
        else hipLaunchKernelGGL(mx6, dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (v4f*)dc, (const int*)dsa, (const int*)dsb);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(D.data(), dc, 1024, hipMemcpyDeviceToHost);
    };
    // ---- stage 1: all ones (e2m3 1.0 in every element of both operands) -> 128
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { put6(&A[l * 32], j, 0x08); put6(&B[l * 32], j, 0x08); }
    run(0);
    printf("stage 1 (all ones, fp6 x fp6): D[lane 0] = %g %g %g %g, D[lane 37] = %g %g %g %g\n", D[0], D[1], D[2], D[3], D[148], D[149], D[150], D[151]);
    // ---- stage 2: scale byte 0 of lane 21 (A) doubled
    SA[21] = 0x7f7f7f80;
    run(0);
    printf("stage 2 (lane 21: A scale byte 0 = 2.0): outputs != 128:");
    for (int i = 0; i < 256; ++i) if (D[i] != 128.f) printf(" [lane %d r %d]=%g", i >> 2, i & 3, D[i]);
    printf("\n");
    SA[21] = 0x7f7f7f7f;
    // ---- stage 3: one-hot scan
    std::vector<uint8_t> cand(128 * 64 * 32, 0);
    for (int x = 0; x < 128; ++x) put6(&cand[(x * 64 + 16 * (x >> 5)) * 32], x & 31, 0x08);
    (void)hipMalloc(&dcand, cand.size());
    (void)hipMemcpy(dcand, cand.data(), cand.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(scan6, dim3(1), dim3(64), 0, 0, (const v8i*)dcand, (const v8i*)dcand, (float*)dscan);
    (void)hipDeviceSynchronize();
    std::vector<float> S(128 * 128);
    (void)hipMemcpy(S.data(), dscan, 128 * 128 * 4, hipMemcpyDeviceToHost);
    int diag = 0, total = 0, nans = 0, shown = 0;
    for (int x = 0; x < 128; ++x)
        for (int y = 0; y < 128; ++y) {
            const float v = S[x * 128 + y];
            if (v != v) { ++nans; continue; }
            if (v != 0.f) { ++total; diag += (x == y); if (x != y && shown++ < 24) printf("   A(g %d, elem %2d) <-> B(g %d, elem %2d)  value %g\n", x >> 5, x & 31, y >> 5, y & 31, v); }
        }
    printf("stage 3 (one-hot, element j = bits 6j..6j+5): %d nonzero pairs, %d on the diagonal (same lane group, same element), %d NaNs\n", total, diag, nans);
    // ---- stage 4: random integers + random scales under: lane l = (row | col l & 15, k = 32 (l >> 4) + element j); scale of block kb from lane (l & 15) + 16 kb, byte 0
    srand(4242);
    std::vector<int> Ai(2048), Bi(2048);
    for (int mixed = 0; mixed < 2; ++mixed)
        for (int mode = 0; mode < 3; ++mode) {          // 0: random data, unit scales; 1: ones, random scales; 2: both
            std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 32; ++j) {
                    Ai[l * 32 + j] = mode == 1 ? 1 : rand() % 9 - 4; Bi[l * 32 + j] = mode == 1 ? 1 : rand() % 9 - 4;
                    put6(&A[l * 32], j, e2m3_of_int(Ai[l * 32 + j]));
                    if (mixed) B[l * 32 + j] = e4m3_of_int(Bi[l * 32 + j]); else put6(&B[l * 32], j, e2m3_of_int(Bi[l * 32 + j]));
                }
            for (int l = 0; l < 64; ++l) {
                SA[l] = 0x7f7f7f00u | (uint32_t)(mode == 0 ? 127 : 126 + rand() % 4);
                SB[l] = 0x7f7f7f00u | (uint32_t)(mode == 0 ? 127 : 126 + rand() % 4);
            }
            run(mixed);
            for (int hyp = 0; hyp < 2; ++hyp) {      // B (e4m3 in the mixed case): hyp 0 = k = 32 g + byte; hyp 1 = the fp8 x fp8 layout (bytes 0-15: k 16 g.., bytes 16-31: k 64 + 16 g..)
                double err = 0, mag = 0;
                for (int row = 0; row < 16; ++row)
                    for (int col = 0; col < 16; ++col) {
                        double s = 0;
                        for (int k = 0; k < 128; ++k) {
                            const int ga = k >> 5, ja = k & 31;
                            int gb = k >> 5, jb = k & 31;
                            if (mixed && hyp == 1) { gb = (k & 63) >> 4; jb = (k & 15) + 16 * (k >> 6); }
                            const int kb = k >> 5;
                            const double sca = ldexp(1.0, (int)(SA[row + 16 * kb] & 255) - 127), scb = ldexp(1.0, (int)(SB[col + 16 * kb] & 255) - 127);
                            s += sca * scb * Ai[(row + 16 * ga) * 32 + ja] * Bi[(col + 16 * gb) * 32 + jb];
                        }
                        const double d = D[(col + 16 * (row >> 2)) * 4 + (row & 3)];
                        err += fabs(d - s); mag += fabs(s);
                    }
                printf("stage 4 %s mode %d B-layout hyp %d: total |D - ref| = %g of %g\n", mixed ? "fp6 x fp8" : "fp6 x fp6", mode, hyp, err, mag);
                if (!mixed) break;
            }
        }
    // ---- rate
    std::vector<int> h(65536);
    srand(1);
    for (auto& v : h) {      // exponent fields away from NaN / inf for every interpretation (f16 halves, e4m3 bytes; every 6-bit pattern is a finite e2m3)
        unsigned x = 0;
        for (int b = 0; b < 4; ++b) x |= (unsigned)((rand() & 0x80) | (0x20 + (rand() % 0x30))) << (8 * b);
        v = (int)x;
    }
    int* d; float* o; unsigned long long* c;
    (void)hipMalloc(&d, 65536 * 4); (void)hipMalloc(&o, 256 * 8 * 256 * 4); (void)hipMalloc(&c, 4096 * 16);
    (void)hipMemcpy(d, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[8] = {"f16 16x16x32            ", "MX e4m3 16x16x128       ", "MX e2m3 16x16x128       ", "GEMM mix 2 f16 + 1 e4m3 ", "GEMM mix 2 f16 + 1 e2m3 ",
                            "MX A e2m3 x B e4m3      ", "MX A e4m3 x B e2m3      ", "GEMM mix 2 f16 + 1 (A e2m3 x B e4m3)"};
    for (int wgs : {256, 512}) {
        for (int kind = 0; kind < 8; ++kind) {
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                if (kind == 3) hipLaunchKernelGGL(rate<3>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                if (kind == 4) hipLaunchKernelGGL(rate<4>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                if (kind == 5) hipLaunchKernelGGL(rate<5>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                if (kind == 6) hipLaunchKernelGGL(rate<6>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                if (kind == 7) hipLaunchKernelGGL(rate<7>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> hc(2 * wgs);
                (void)hipMemcpy(hc.data(), c, 16 * wgs, hipMemcpyDeviceToHost);
                const double cyc = (double)hc[0], us = (double)hc[1] / 100.0;
                // per iteration and wave: 8 (i, j) pairs; kinds 0: 16 f16 MFMAs; 1 / 2: 8 scaled; 3 / 4: 16 f16 + 8 scaled = 64 K of the fp32 mode's product per pair
                if (rep == 2) printf("%3d WGs x 4 waves  %s: %8.3f ms  clock %.2f GHz  cycles per (i, j) pair %.1f\n", wgs, names[kind], ms, cyc / us / 1e3, cyc / iters / 8 / (wgs == 512 ? 2 : 1));
            }
        }
    }
    return 0;
}
