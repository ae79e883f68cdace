// Probe (GPU box): operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (cbsz = blgp = 0), found with exact data.
//   hipcc --offload-arch=gfx950 -O2 tools/mb_mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
// Stage 1: all ones -> 128.  Stage 2: one lane's scale byte doubled -> which outputs move.  Stage 3: one-hot A element against one-hot B
// element -> which (lane, byte) pairs share a k.  Stage 4: random integers + random scales against layout hypotheses.
// RESULT (profiles/r04/mx_mfma_layout_probe.txt): lane l = (row | col l & 15, group g = l >> 4) holds bytes 0-15 = k 16 g .. 16 g + 15 and bytes
// 16-31 = k 64 + 16 g .. (two 16-byte chunks g and 4 + g of a K-contiguous 128-byte row: the access pattern of the f16 fragments fa[0] / fa[1]
// of this library's panels); the E8M0 scale of the 32-k block kb of that row | col is taken from lane (l & 15) + 16 kb, byte `opsel` of the scale
// VGPR; D as every 16x16 MFMA (row = 4 (l >> 4) + r, col = l & 15).  e4m3 = OCP (0x38 = 1.0).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OPA, int OPB>
__global__ void mx(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, OPA, sa[l], OPB, sb[l]);
    c[l] = acc;
}

// one-hot scan: candidate x = (group ga, byte ja): A has 1.0 in lane 16 ga, byte ja (row 0); candidate y = (gb, jb): B has 1.0 in lane 16 gb,
// byte jb (col 0).  The operand images come from the host (no dynamic register indexing here).  out[x][y] = D[0][0].
__global__ void scan(const v8i* acand, const v8i* bcand, float* out) {
    const int l = threadIdx.x;
    int one = 0x7f7f7f7f;
    asm volatile("" : "+v"(one));
    for (int x = 0; x < 128; ++x) {
        const v8i a = acand[x * 64 + l];
        for (int y = 0; y < 128; ++y) {
            const v8i b = bcand[y * 64 + l];
            v4f acc = {0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, one, 0, one);
            if (l == 0) out[x * 128 + y] = acc[0];
        }
    }
}

static uint8_t e4m3_of_int(int v) {      // v in -4..4
    static const uint8_t pos[5] = {0x00, 0x38, 0x40, 0x44, 0x48};      // 0, 1, 2, 3, 4
    return v < 0 ? (uint8_t)(pos[-v] | 0x80) : pos[v];
}

int main() {
    void *da, *db, *dc, *dsa, *dsb, *dscan;
    (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dc, 64 * 16); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256);
    (void)hipMalloc(&dscan, 128 * 128 * 4);
    std::vector<uint8_t> A(2048, 0x38), B(2048, 0x38);
    std::vector<uint32_t> SA(64, 0x7f7f7f7f), SB(64, 0x7f7f7f7f);
    std::vector<float> D(256);
    auto run = [&](int op) {
        (void)hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
        (void)hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice);
        if (op == 0) hipLaunchKernelGGL((mx<0, 0>), dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (v4f*)dc, (const int*)dsa, (const int*)dsb);
        if (op == 1) hipLaunchKernelGGL((mx<1, 2>), dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (v4f*)dc, (const int*)dsa, (const int*)dsb);
        if (op == 2) hipLaunchKernelGGL((mx<2, 3>), dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (v4f*)dc, (const int*)dsa, (const int*)dsb);
        if (op == 3) hipLaunchKernelGGL((mx<3, 1>), dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (v4f*)dc, (const int*)dsa, (const int*)dsb);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(D.data(), dc, 1024, hipMemcpyDeviceToHost);
    };
    // ---- stage 1
    run(0);
    printf("stage 1 (all ones): D[lane 0] = %g %g %g %g, D[lane 37] = %g %g %g %g\n", D[0], D[1], D[2], D[3], D[148], D[149], D[150], D[151]);
    // ---- stage 2: A scale byte 0 of lane 21 doubled (e = 128)
    SA[21] = 0x7f7f7f80;
    run(0);
    printf("stage 2 (lane 21: A scale byte 0 = 2.0): outputs != 128:");
    for (int i = 0; i < 256; ++i) if (D[i] != 128.f) printf(" [lane %d r %d]=%g", i >> 2, i & 3, D[i]);
    printf("\n");
    SA[21] = 0x7f7f7f7f;
    SB[38] = 0x7f7f807f;     // B scale byte 1 of lane 38, opsel_b = ... try op 0 (byte 0: no change expected) and op with OPB = 1
    run(0);
    int moved = 0; for (int i = 0; i < 256; ++i) moved += D[i] != 128.f;
    printf("stage 2b (lane 38: B scale BYTE 1 = 2.0, opsel 0): %d outputs moved (expect 0 if opsel picks the byte)\n", moved);
    run(3);                   // OPB = 1
    printf("stage 2c (same, opsel_b = 1): outputs != 128:");
    for (int i = 0; i < 256; ++i) if (D[i] != 128.f) printf(" [lane %d r %d]=%g", i >> 2, i & 3, D[i]);
    printf("\n");
    SB[38] = 0x7f7f7f7f;
    // ---- stage 3
    std::vector<uint8_t> cand(128 * 64 * 32, 0);
    for (int x = 0; x < 128; ++x) cand[(x * 64 + 16 * (x >> 5)) * 32 + (x & 31)] = 0x38;
    void* dcand;
    (void)hipMalloc(&dcand, cand.size());
    (void)hipMemcpy(dcand, cand.data(), cand.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(scan, dim3(1), dim3(64), 0, 0, (const v8i*)dcand, (const v8i*)dcand, (float*)dscan);
    (void)hipDeviceSynchronize();
    std::vector<float> S(128 * 128);
    (void)hipMemcpy(S.data(), dscan, 128 * 128 * 4, hipMemcpyDeviceToHost);
    printf("stage 3: A element (lane group ga, byte ja) meets B element (lane group gb, byte jb):\n");
    int diag = 0, total = 0, nans = 0, shown = 0;
    for (int x = 0; x < 128; ++x)
        for (int y = 0; y < 128; ++y) {
            const float v = S[x * 128 + y];
            if (v != v) { ++nans; continue; }
            if (v != 0.f) { ++total; diag += (x == y); if (x != y && shown++ < 40) printf("   A(g %d, byte %2d) <-> B(g %d, byte %2d)  value %g\n", x >> 5, x & 31, y >> 5, y & 31, v); }
        }
    printf("   %d nonzero pairs, %d of them on the diagonal (same group, same byte), %d NaNs\n", total, diag, nans);
    // ---- stage 4: random integers and / or random scales under: lane l = (row | col l & 15, k = 32 (l >> 4) + byte), scale byte of lane l covers its 32 bytes
    srand(777);
    std::vector<int> Ai(2048), Bi(2048);
    const int opa[4] = {0, 1, 2, 3}, opb[4] = {0, 2, 3, 1};
    for (int mode = 0; mode < 3; ++mode) {          // 0: random data, unit scales; 1: ones, random scales; 2: both
        for (int i = 0; i < 2048; ++i) {
            Ai[i] = mode == 1 ? 1 : rand() % 9 - 4; Bi[i] = mode == 1 ? 1 : rand() % 9 - 4;
            A[i] = e4m3_of_int(Ai[i]); B[i] = e4m3_of_int(Bi[i]);
        }
        for (int l = 0; l < 64; ++l) {
            SA[l] = SB[l] = 0;
            for (int by = 0; by < 4; ++by) {
                SA[l] |= (uint32_t)(mode == 0 ? 127 : 126 + rand() % 4) << (8 * by);
                SB[l] |= (uint32_t)(mode == 0 ? 127 : 126 + rand() % 4) << (8 * by);
            }
        }
        for (int op = 0; op < 4; ++op) for (int hyp = 0; hyp < 4; ++hyp) {
            run(op);
            double err = 0, mag = 0;
            int worst = -1; double werr = 0, wref = 0;
            for (int row = 0; row < 16; ++row)
                for (int col = 0; col < 16; ++col) {
                    // hypothesis hyp: a lane's 32 bytes are 32 / blk pieces of blk contiguous k; piece p of lane group g is k = 4 blk p + blk g ...;
                    // the scale of the 32-k block kb comes from lane (row | col) + 16 kb
                    double s = 0;
                    const int blk = 32 >> hyp;
                    for (int g = 0; g < 4; ++g) {
                        const int la = row + 16 * g, lb = col + 16 * g;
                        for (int j = 0; j < 32; ++j) {
                            const int k = blk * g + (j % blk) + 4 * blk * (j / blk);
                            const int kb = k >> 5;
                            const double sca = ldexp(1.0, (int)((SA[row + 16 * kb] >> (8 * opa[op])) & 255) - 127), scb = ldexp(1.0, (int)((SB[col + 16 * kb] >> (8 * opb[op])) & 255) - 127);
                            s += sca * scb * Ai[la * 32 + j] * Bi[lb * 32 + j];
                        }
                    }
                    const double d = D[(col + 16 * (row >> 2)) * 4 + (row & 3)];
                    err += fabs(d - s); mag += fabs(s);
                    if (fabs(d - s) > werr) { werr = fabs(d - s); worst = row * 16 + col; wref = s; }
                }
            printf("stage 4 mode %d hyp blk %2d opsel (%d,%d): total |D - ref| = %g of %g; worst at (row %d, col %d): got %g want %g\n", mode, 32 >> hyp, opa[op], opb[op], err, mag,
                   worst >> 4, worst & 15, worst < 0 ? 0.0 : D[((worst & 15) + 16 * ((worst >> 4) >> 2)) * 4 + ((worst >> 4) & 3)], wref);
        }
    }
    return 0;
}
