// radzero_hip — common device helpers for gfx950 (CDNA4) kernels.
//
// Conventions used by every MFMA kernel in this library:
//  * Operand "fragment" = 8 K-contiguous elements per lane for one row/column (lane&15) of a 16-wide
//    tile, K sub-block (lane>>4).  For 16-bit types that is exactly the v_mfma_f32_16x16x32 operand
//    (16 B per lane); for f32 the same 8 elements (32 B per lane) feed 8 x v_mfma_f32_16x16x4_f32,
//    element j of every lane forming one K=4 step.  A and B fragments have identical layout, so
//    mma(a, b) computes D[row of a][col of b] and mma(b, a) its transpose.
//  * D layout (all dtypes): lane holds D[row = 4*(lane>>4) + r][col = lane&15], r = 0..3.
//  * LDS tiles are "panels" of [rows][128 bytes]; the 16-byte chunk c of row r is stored at chunk
//    position c ^ ((r>>1)&7).  Two 128-B rows share one 256-B bank row, so this XOR makes the
//    16 rows x 1 chunk pattern of an MFMA fragment read (ds_read_b128 / ds_read_b64) conflict-free.
//    Tiles are filled by global_load_lds_dwordx4 (LDS image is lane-linear, the XOR is applied to the
//    per-lane SOURCE address) and read back with the same XOR.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rz {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

template <typename T> struct Traits;
template <> struct Traits<float> {
    typedef f32x8 frag;
    typedef f32x4 vec4;
    static constexpr int kDType = DT_F32;
};
template <> struct Traits<bf16_t> {
    typedef bf16x8 frag;
    typedef bf16x4 vec4;
    static constexpr int kDType = DT_BF16;
};
template <> struct Traits<f16_t> {
    typedef f16x8 frag;
    typedef f16x4 vec4;
    static constexpr int kDType = DT_F16;
};

// elements per 128-byte LDS panel row
template <typename T> __host__ __device__ constexpr int panel_elems() { return 128 / (int)sizeof(T); }

// ---- MFMA wrappers: acc += a (rows) x b (cols) over the fragment's 32 K-elements ----
__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const f16x8& a, const f16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const f32x8& a, const f32x8& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
}

// ---- conversions ----
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) { return (f16_t)x; }
template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }

template <typename T> __device__ __forceinline__ typename Traits<T>::vec4 pack4(float a, float b, float c, float d) {
    typename Traits<T>::vec4 v;
    v[0] = from_f32<T>(a); v[1] = from_f32<T>(b); v[2] = from_f32<T>(c); v[3] = from_f32<T>(d);
    return v;
}

// Output "type" of the fp32 mode's hi/lo-split path: a value x leaves an epilogue as f16 planes hi = f16(x), lo = f16(x - hi).
struct split_f16 {};
// Overflow guard of that path: the planes are f16, so a value beyond +-65504 (or a NaN) would become inf where fp32 stays finite.  Every
// producer of planes ORs 1 into `*flag` when it meets one (integer compare on the magnitude bits: also true for NaN, and immune to
// -fno-honor-nans); api.hip reads the word once per forward and repeats that forward on the exact-fp32 kernels.  Values below 6e-5 lose
// (part of) their lo plane to f16's subnormal range: an ABSOLUTE error <= 2^-25 per element, far inside the 1e-3 parity bar.
__device__ __forceinline__ void flag_f16_range(const f32x4& v, unsigned* flag) {
    const unsigned a = max(max(__float_as_uint(v[0]) & 0x7fffffffu, __float_as_uint(v[1]) & 0x7fffffffu),
                           max(__float_as_uint(v[2]) & 0x7fffffffu, __float_as_uint(v[3]) & 0x7fffffffu));
    if (a > 0x477fe000u && flag) atomicOr(flag, 1u);          // 0x477fe000 = 65504.0f
}
__device__ __forceinline__ void split4(const f32x4& v, f16x4& hi, f16x4& lo, unsigned* flag = nullptr) {
    flag_f16_range(v, flag);
    hi = pack4<f16_t>(v[0], v[1], v[2], v[3]);
    float m1 = -1.0f;                // opaque: v - f16(v) as one v_fma_mix_f32 on the packed half (hipcc would split it into convert + subtract)
    asm("" : "+s"(m1));
    lo = pack4<f16_t>(__builtin_fmaf((float)hi[0], m1, v[0]), __builtin_fmaf((float)hi[1], m1, v[1]), __builtin_fmaf((float)hi[2], m1, v[2]),
                      __builtin_fmaf((float)hi[3], m1, v[3]));
}

// ---- fp32 mode, MX form (round 4): a product a b = a_hi b_hi + (a_lo b_hi + a_hi b_lo); the two correction terms sit 2^-11 below the main
// one and need ~4 significant bits, so they run as ONE block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands, 2 x the
// f16 rate) over K' = 128 = [a_lo(64) | a_hi8(64)] . [b_hi8(64) | b_lo(64)] beside the two f16 MFMAs of a_hi b_hi: 4 MFMA-units per 64 K
// instead of 6.  An operand row is [hi plane: K f16][K / 64 groups of 128 bytes: activations lo8 x 64 | hi8 x 64, weights hi8 x 64 | lo8 x 64]
// = 4 K bytes (6 K in the three-plane f16 form).  Scales are FIXED powers of two per plane kind (E8M0 bytes below) — no per-block scale
// plumbing: e4m3 spans 2^-9 .. 448 times its scale, an element far below its plane's scale loses RELATIVE precision only where its
// ABSOLUTE contribution is negligible (the correction terms are already 2^-11 of the product), and an element above it is clamped and
// flagged (the forward is then repeated on the exact kernels, as for the f16 planes).  Operand / scale layout of the instruction:
// tools/mb_mx_probe.hip (profiles/r04/mx_mfma_layout_probe.txt); conversion numerics (RNE, overflow -> NaN, hence the clamp):
// tools/mb_fp8_cvt.hip.
struct split_mx {};       // output "type": the next GEMM's A operand in the form above
struct split_mxa {};      // output "type": the attention's q | k / V^T operands: hi f16 plane + e4m3 pair plane (attention.hip "MXA")
constexpr float MX_A_HI_SCALE = 4.0f, MX_A_LO_SCALE = 4.0f / 2048.0f;          // activations: hi8 covers |x| <= 1792, lo8 = (x - f16(x)) / 2^-9
constexpr float MX_W_HI_SCALE = 1.0f / 16.0f, MX_W_LO_SCALE = 1.0f / 32768.0f; // weights: hi8 covers |w| <= 28
constexpr int MX_E8_A_HI = 129, MX_E8_A_LO = 118, MX_E8_W_HI = 123, MX_E8_W_LO = 112;      // E8M0 = 127 + log2(scale)
__device__ __forceinline__ uint32_t cvt4_e4m3(float a, float b, float c, float d) {
    int w = __builtin_bit_cast(int, a);          // "old" word of the first (tied) convert: a value that dies here, not a fresh zero (a v_mov per word)
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
// the same through gfx950's scaled convert: v_cvt_scalef32_pk_fp8_f32 DIVIDES by the power of two of its scale operand (the mantissa is ignored),
// rounds to nearest even, and gives NaN beyond +-448 like the plain convert (tools/mb_fp8_cvt_scale.hip) — e4m3(x / scale) without the multiply
constexpr float MX_CVT_SCALE_2P11 = 1.0f / 2048.0f;
__device__ __forceinline__ uint32_t cvt4_e4m3_scaled(float a, float b, float c, float d, float scale) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 w = __builtin_bit_cast(s16x2, a);      // "old" word of the first (tied) convert: a value that dies here, not a fresh zero
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, a, b, scale, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, c, d, scale, true);
    return __builtin_bit_cast(uint32_t, w);
}
// x - f16(x) from the packed half itself: one v_fma_mix_f32 (the -1 is kept opaque, or hipcc turns the fma back into convert + subtract)
__device__ __forceinline__ float opaque_minus_one() { float m = -1.0f; asm("" : "+s"(m)); return m; }
__device__ __forceinline__ float minus_f16(float x, f16_t h, float m1) { return __builtin_fmaf((float)h, m1, x); }
__device__ __forceinline__ float clamp_abs(float x, float lim) { return fminf(fmaxf(x, -lim), lim); }
__device__ __forceinline__ float clamp448(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
__device__ __forceinline__ void pair4_mx(const f32x4& v, const f16x4& hi, uint32_t& lo8, uint32_t& hi8, float hi_scale, float lo_scale);
// hi = f16(v); lo8 = e4m3((v - hi) / lo_scale); hi8 = e4m3(clamp(v / hi_scale)); flags |v| beyond the hi8 range (or the f16 range, or NaN)
__device__ __forceinline__ void split4_mx(const f32x4& v, f16x4& hi, uint32_t& lo8, uint32_t& hi8, float hi_scale, float lo_scale, unsigned* flag) {
    const float lim = 448.f * hi_scale;
    const unsigned a = max(max(__float_as_uint(v[0]) & 0x7fffffffu, __float_as_uint(v[1]) & 0x7fffffffu),
                           max(__float_as_uint(v[2]) & 0x7fffffffu, __float_as_uint(v[3]) & 0x7fffffffu));
    if (a > __float_as_uint(fminf(lim, 65504.f)) && flag) atomicOr(flag, 1u);
    hi = pack4<f16_t>(v[0], v[1], v[2], v[3]);
    pair4_mx(v, hi, lo8, hi8, hi_scale, lo_scale);
}
// the two e4m3 words alone (hi = f16(v) already known, range already flagged): 3 + 2 vector instructions per value (fma_mix, clamp, half a scaled
// convert; clamp, half a scaled convert)
__device__ __forceinline__ void pair4_mx(const f32x4& v, const f16x4& hi, uint32_t& lo8, uint32_t& hi8, float hi_scale, float lo_scale) {
    const float m1 = opaque_minus_one();
    const float ll = 448.f * lo_scale, lh = 448.f * hi_scale;
    lo8 = cvt4_e4m3_scaled(clamp_abs(minus_f16(v[0], hi[0], m1), ll), clamp_abs(minus_f16(v[1], hi[1], m1), ll), clamp_abs(minus_f16(v[2], hi[2], m1), ll),
                           clamp_abs(minus_f16(v[3], hi[3], m1), ll), lo_scale);
    hi8 = cvt4_e4m3_scaled(clamp_abs(v[0], lh), clamp_abs(v[1], lh), clamp_abs(v[2], lh), clamp_abs(v[3], lh), hi_scale);
}
__device__ __forceinline__ void flag_mx_range(const f32x4& v, float hi_scale, unsigned* flag) {
    const unsigned a = max(max(__float_as_uint(v[0]) & 0x7fffffffu, __float_as_uint(v[1]) & 0x7fffffffu),
                           max(__float_as_uint(v[2]) & 0x7fffffffu, __float_as_uint(v[3]) & 0x7fffffffu));
    if (a > __float_as_uint(fminf(448.f * hi_scale, 65504.f)) && flag) atomicOr(flag, 1u);
}
// byte offsets inside an MX operand row of K elements (row = 4 K bytes): element k
__device__ __forceinline__ int64_t mx_pair_off(int K, int k) { return (int64_t)2 * K + (k >> 6) * 128 + (k & 63); }      // first byte plane of its 64-group; second = + 64
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
// acc += sum over the 128-byte K' tile: a = chunks (lg, 4 + lg) of the A row, b likewise of the W row; sa / sb: the lane's E8M0 scale bytes
__device__ __forceinline__ f32x4 mma_mx(const f16x8& a0, const f16x8& a1, const f16x8& b0, const f16x8& b1, f32x4 c, int sa, int sb) {
    const i32x4 x0 = __builtin_bit_cast(i32x4, a0), x1 = __builtin_bit_cast(i32x4, a1), y0 = __builtin_bit_cast(i32x4, b0), y1 = __builtin_bit_cast(i32x4, b1);
    const i32x8 a = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7), b = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
}

// 8 floats -> one operand fragment (pairwise packed converts: v_cvt_pk_bf16_f32 / v_cvt_f16 + pack)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ typename Traits<T>::frag pack8(float a, float b, float c, float d, float e, float f, float g, float h);
template <> __device__ __forceinline__ f32x8 pack8<float>(float a, float b, float c, float d, float e, float f, float g, float h) {
    return (f32x8){a, b, c, d, e, f, g, h};
}
template <> __device__ __forceinline__ bf16x8 pack8<bf16_t>(float a, float b, float c, float d, float e, float f, float g, float h) {
    const bf16x2 p0 = __builtin_convertvector((f32x2){a, b}, bf16x2), p1 = __builtin_convertvector((f32x2){c, d}, bf16x2);
    const bf16x2 p2 = __builtin_convertvector((f32x2){e, f}, bf16x2), p3 = __builtin_convertvector((f32x2){g, h}, bf16x2);
    return __builtin_shufflevector(__builtin_shufflevector(p0, p1, 0, 1, 2, 3), __builtin_shufflevector(p2, p3, 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
}
template <> __device__ __forceinline__ f16x8 pack8<f16_t>(float a, float b, float c, float d, float e, float f, float g, float h) {
    const f16x2 p0 = __builtin_convertvector((f32x2){a, b}, f16x2), p1 = __builtin_convertvector((f32x2){c, d}, f16x2);
    const f16x2 p2 = __builtin_convertvector((f32x2){e, f}, f16x2), p3 = __builtin_convertvector((f32x2){g, h}, f16x2);
    return __builtin_shufflevector(__builtin_shufflevector(p0, p1, 0, 1, 2, 3), __builtin_shufflevector(p2, p3, 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- LDS panel addressing ----
// byte offset of 16-B chunk `c` (0..7) of row `r` inside a [rows][128 B] panel
__device__ __forceinline__ int panel_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// One wave fills 8 consecutive panel rows (1 KiB) with one global_load_lds_dwordx4.
//  lds_wave_base : wave-uniform LDS address of panel row `row8` (multiple of 8)
//  gsrc_row0     : global address of (tile row 0, panel byte 0); row stride `ld_bytes`
// Lane l lands at LDS row row8 + (l>>3), chunk position (l&7); it must therefore FETCH global chunk
// (l&7) ^ ((row>>1)&7) of that row.
// SWZ selects the 3-bit XOR term of the row: 0 = standard ((r>>1)&7), 1 = attention K tile (swz_k below).
__device__ __forceinline__ int swz_std(int r) { return (r >> 1) & 7; }
// K tile of the flash kernel: one fragment read touches rows {8a + 4t + b : a,b in 0..3} (t fixed), see attention.hip
__device__ __forceinline__ int swz_k(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }
template <int SWZ = 0>
__device__ __forceinline__ void glds_rows8(char* lds_wave_base, const char* gsrc_row0, int64_t ld_bytes, int row8, int lane) {
    const int r = row8 + (lane >> 3);
    const int c = (lane & 7) ^ (SWZ == 0 ? swz_std(r) : swz_k(r));
    const char* src = gsrc_row0 + (int64_t)r * ld_bytes + (c << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Read one fragment (row = r, K sub-block kb of width 8 elements) from a panel.
// 16-bit: chunk index = kb (8 elem = 16 B).  f32: chunks 2kb, 2kb+1.
template <typename T> __device__ __forceinline__ typename Traits<T>::frag lds_frag(const char* panel, int r, int kb);
template <> __device__ __forceinline__ bf16x8 lds_frag<bf16_t>(const char* panel, int r, int kb) {
    return *reinterpret_cast<const bf16x8*>(panel + panel_off(r, kb));
}
template <> __device__ __forceinline__ f16x8 lds_frag<f16_t>(const char* panel, int r, int kb) {
    return *reinterpret_cast<const f16x8*>(panel + panel_off(r, kb));
}
template <> __device__ __forceinline__ f32x8 lds_frag<float>(const char* panel, int r, int kb) {
    f32x4 lo = *reinterpret_cast<const f32x4*>(panel + panel_off(r, 2 * kb));
    f32x4 hi = *reinterpret_cast<const f32x4*>(panel + panel_off(r, 2 * kb + 1));
    f32x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return v;
}

// ---- wave reductions (64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf in fp32, branch-free, <= 1 ulp (N. Juffa's two-range minimax fit: |a| <= 0.9277 a + a P(a^2), beyond 1 - exp(Q(|a|)); both evaluated, one
// selected; max relative error 3.2e-8 in exact arithmetic, checked against scipy over [-6, 6] and 1e-8 .. 10).  libm's erff is a branchy
// 60-instruction body: inlined 128 x in a GEMM epilogue it made hipcc spill 270-570 VGPRs (round 4), and the fp32 mode does not need it.
__device__ __forceinline__ float erf_f32(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_f32(x * 0.70710678118654752440f)); }

// GELU for the 16-bit modes, transcendental-free:  gelu(x) = max(x, 0) - a Q(a),  a = min(|x|, 4.5),
// Q(a) = (1 - erf(a / sqrt 2)) / 2 replaced by its degree-9 Chebyshev fit on [0, 4.5] evaluated by Horner in
// u = 2a/4.5 - 1 (well conditioned in fp32).  |error| <= 2.3e-5 absolute, <= 3.2e-5 relative near 0 (simulated in fp32
// over [-12, 12] against scipy's erf) — 40x below one f16 ulp at 1 and 350x below one bf16 ulp — at 9 VALU
// issue slots per element instead of 14 for the rcp + exp2 form of Abramowitz-Stegun 7.1.26 (the fc1 epilogue is
// VALU-bound: 65536 elements per 256x256 tile).  fp32 mode uses libm erff (gelu_erf above).
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float a = fminf(fabsf(x), 4.5f);
    const float u = fmaf(a, 2.0f / 4.5f, -1.0f);
    float p = fmaf(u, -1.913512412e-02f, 2.984178795e-02f);
    p = fmaf(p, u, 4.896834610e-02f);
    p = fmaf(p, u, -1.296090571e-01f);
    p = fmaf(p, u, 3.827712359e-02f);
    p = fmaf(p, u, 1.566413223e-01f);
    p = fmaf(p, u, -2.468006774e-01f);
    p = fmaf(p, u, 1.809132537e-01f);
    p = fmaf(p, u, -7.131592906e-02f);
    p = fmaf(p, u, 1.222076767e-02f);
    return fmaxf(x, 0.f) - a * p;
}
template <typename T> __device__ __forceinline__ float gelu_for(float x) {
    if constexpr (sizeof(T) == 4) return gelu_erf(x);      // fp32 parity mode: libm erff
    else return gelu_erf_fast(x);
}

// XCD-aware remap of a linear workgroup id: consecutive `group`-sized runs of logical ids land on one
// XCD (blocks b and b+8 share an XCD under round-robin dispatch).  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

}  // namespace rz
