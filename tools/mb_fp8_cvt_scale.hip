// Probe (GPU box): semantics of gfx950's v_cvt_scalef32_pk_fp8_f32 / _f16 (convert WITH a power-of-two scale operand): is the result
// e4m3(x / scale) or e4m3(x * scale), how does it round, what happens beyond +-448?
//   hipcc --offload-arch=gfx950 -O2 tools/mb_fp8_cvt_scale.hip -o /tmp/fp8_cvts && /tmp/fp8_cvts
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef short v2s __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, int* o, int n, float sc) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    v2s r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[l], 0.f, sc, false);                  // byte 0: from f32
    v2h h = {(_Float16)x[l], (_Float16)0.f};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, h, sc, true);                             // byte 2: from f16
    o[l] = __builtin_bit_cast(int, r);
}
static double dec(uint8_t b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    double v = e == 0 ? ldexp((double)m, 1 - 7 - 3) : ldexp(1.0 + m / 8.0, e - 7);
    return s ? -v : v;
}
static uint8_t enc_rne(double x) {
    uint8_t best = 0; double bd = 1e300;
    for (int b = 0; b < 256; ++b) {
        if ((b & 0x7f) == 0x7f) continue;
        const double d = fabs(dec((uint8_t)b) - x);
        if (d < bd || (d == bd && !(b & 1))) { bd = d; best = (uint8_t)b; }
    }
    return best;
}
int main() {
    std::vector<float> xs;
    for (int e = -24; e <= 12; ++e) for (int m = 0; m < 64; ++m) { const float v = ldexpf(1.f + m / 64.f, e); xs.push_back(v); xs.push_back(-v); }
    const int n = (int)xs.size();
    float* dx; int* dо;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dо, n * 4);
    (void)hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<int> o(n);
    const float scales[6] = {1.f, 2048.f, 1.f / 2048.f, 4.f, 0.25f, 3.0f};
    for (float sc : scales) {
        hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dо, n, sc);
        (void)hipMemcpy(o.data(), dо, n * 4, hipMemcpyDeviceToHost);
        const double s2 = ldexp(1.0, (int)floor(log2((double)sc)));          // the scale's power of two (3.0 -> 2: is the mantissa ignored?)
        for (int src = 0; src < 2; ++src) {
            int div_ok = 0, mul_ok = 0, tot = 0, sat = 0, nan = 0, over = 0;
            for (int i = 0; i < n; ++i) {
                const double x = src ? (double)(float)(_Float16)xs[i] : (double)xs[i];
                if (src && (fabs(xs[i]) > 65504 || fabs(xs[i]) < 6.2e-5)) continue;
                const uint8_t g = (o[i] >> (16 * src)) & 0xff;
                const double d = x / s2, m = x * s2;
                if (fabs(d) > 464) { ++over; sat += (g & 0x7f) == 0x7e; nan += (g & 0x7f) == 0x7f; continue; }
                ++tot;
                div_ok += dec(g) == dec(enc_rne(d));
                if (fabs(m) <= 464) mul_ok += dec(g) == dec(enc_rne(m));
            }
            printf("scale %g (%s source): %d in-range values: matches e4m3_rne(x / 2^floor(log2 scale)) %d, e4m3_rne(x * ...) %d; beyond 464 after division: %d, of them saturated to 448: %d, NaN: %d\n",
                   sc, src ? "f16" : "f32", tot, div_ok, mul_ok, over, sat, nan);
        }
    }
    return 0;
}
