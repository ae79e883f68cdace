// Probe (GPU box): numerics of v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 on gfx950 (OCP e4m3fn / e5m2): rounding, subnormals, overflow.
//   hipcc --offload-arch=gfx950 -O2 tools/mb_fp8_cvt.hip -o /tmp/fp8_cvt && /tmp/fp8_cvt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, int* o, int n) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(x[l], 0.f, r, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(x[l], 0.f, r, true);
    o[l] = r;
}
static double dec(uint8_t b, int ebits, int mbits, int bias) {
    const int s = b >> 7, e = (b >> mbits) & ((1 << ebits) - 1), m = b & ((1 << mbits) - 1);
    double v = e == 0 ? ldexp((double)m, 1 - bias - mbits) : ldexp(1.0 + m / (double)(1 << mbits), e - bias);
    return s ? -v : v;
}
static uint8_t enc_rne(double x, int ebits, int mbits, int bias, double maxv) {      // round to nearest even, saturate
    uint8_t best = 0; double bd = 1e300;
    for (int b = 0; b < 256; ++b) {
        if (ebits == 4 && (b & 0x7f) == 0x7f) continue;                              // e4m3fn NaN
        if (ebits == 5 && ((b >> 2) & 31) == 31) continue;                           // e5m2 inf / NaN
        const double v = dec((uint8_t)b, ebits, mbits, bias);
        const double d = fabs(v - x);
        if (d < bd || (d == bd && !(b & 1))) { bd = d; best = (uint8_t)b; }
    }
    (void)maxv;
    return best;
}
int main() {
    std::vector<float> xs;
    for (int e = -14; e <= 17; ++e) for (int m = 0; m < 64; ++m) { const float v = ldexpf(1.f + m / 64.f, e); xs.push_back(v); xs.push_back(-v); }
    xs.push_back(0.f); xs.push_back(448.f); xs.push_back(464.f); xs.push_back(480.f); xs.push_back(1e6f); xs.push_back(57344.f); xs.push_back(61440.f); xs.push_back(65536.f); xs.push_back(1e9f);
    const int n = (int)xs.size();
    float* dx; int* dо;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dо, n * 4);
    (void)hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dо, n);
    std::vector<int> o(n);
    (void)hipMemcpy(o.data(), dо, n * 4, hipMemcpyDeviceToHost);
    int bad8 = 0, badb = 0;
    for (int i = 0; i < n; ++i) {
        const uint8_t f8 = o[i] & 0xff, b8 = (o[i] >> 16) & 0xff;
        const uint8_t w8 = enc_rne(xs[i], 4, 3, 7, 448.0), wb = enc_rne(xs[i], 5, 2, 15, 57344.0);
        const bool in8 = fabs(xs[i]) <= 464.0, inb = fabs(xs[i]) <= 61440.0;
        if (in8 && f8 != w8 && !(dec(f8, 4, 3, 7) == dec(w8, 4, 3, 7))) { if (bad8++ < 8) printf("e4m3 %g: got 0x%02x (%g) want 0x%02x (%g)\n", xs[i], f8, dec(f8, 4, 3, 7), w8, dec(w8, 4, 3, 7)); }
        if (inb && b8 != wb && !(dec(b8, 5, 2, 15) == dec(wb, 5, 2, 15))) { if (badb++ < 8) printf("e5m2 %g: got 0x%02x (%g) want 0x%02x (%g)\n", xs[i], b8, dec(b8, 5, 2, 15), wb, dec(wb, 5, 2, 15)); }
        if (!in8 && i >= n - 9) printf("e4m3 overflow %g -> 0x%02x\n", xs[i], f8);
        if (!inb && i >= n - 9) printf("e5m2 overflow %g -> 0x%02x\n", xs[i], b8);
    }
    printf("%d values: e4m3 mismatches %d, e5m2 mismatches %d (round-to-nearest-even reference, in-range values)\n", n, bad8, badb);
    return 0;
}
