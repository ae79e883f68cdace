// radzero_hip — attention kernels, gfx950.
//
// flash_attn_kernel: softmax(Q K^T) V for the 14 Dinov2 blocks without ever materialising the
//   (B,12,N,N) score tensor (1.36 GB fp32 per image at 1024^2).  Replaces
//   TF:dinov2/modeling_dinov2.py:153-178 (eager_attention_forward) as called from :182-234.
// text_attn_kernel: MPNet self-attention with additive relative-position bias and key-padding mask,
//   TF:mpnet/modeling_mpnet.py:131-171 (bias computed once per forward, :312-348).
//
// Flash kernel data flow (per workgroup = 128 query rows of one (image, head); 4 waves x 32 rows):
//   S^T = K Q^T   : A = K fragment (row = key), B = Q fragment (col = query, held in registers)
//                   -> lane (q = lane&15, g = lane>>4) owns keys 16t+4g+r of every 16-key tile t.
//   softmax       : row statistics are per lane (+2 shuffles across g); no LDS round trip.
//   O^T += V^T P^T: A = V^T fragment (row = d), B = P^T built in registers from the S^T accumulators.
//                   The MFMA's K slot (g, j) is mapped to key 32kk+4g+j (j<4) / 32kk+16+4g+(j-4) for BOTH
//                   operands, so P needs no transpose; V^T rows are key-contiguous because the QKV GEMM
//                   epilogue already wrote V transposed ([B][H][64][Npad]).
//   K and V^T tiles (64 keys) are double-buffered in LDS via global_load_lds_dwordx4 (rz_common.h panels).
#include "rz_common.h"
#include "rz_kernels.h"

namespace rz {

constexpr int FA_QROWS = 128;   // query rows per workgroup
constexpr int FA_KEYS = 64;     // keys per KV tile
constexpr float LOG2E = 1.4426950408889634f;

template <typename T> struct FaCfg {
    static constexpr int NPAN = 64 * (int)sizeof(T) / 128;   // 128-B panels per 64-element row (1 or 2)
    static constexpr int TILE_BYTES = NPAN * 64 * 128;       // one K (or V^T) tile
};

// 4 elements at `key_a` and 4 at `key_b` of row `row` of a [64 rows][64 elems] tile stored as NPAN panels.
template <typename T>
__device__ __forceinline__ typename Traits<T>::frag lds_frag_split(const char* tile, int row, int key_a, int key_b) {
    typedef typename Traits<T>::vec4 v4;
    auto addr = [&](int key) {
        const int byte = key * (int)sizeof(T);
        const int pan = byte >> 7, within = byte & 127;
        return tile + pan * (64 * 128) + panel_off(row, within >> 4) + (within & 15);
    };
    const v4 lo = *reinterpret_cast<const v4*>(addr(key_a));
    const v4 hi = *reinterpret_cast<const v4*>(addr(key_b));
    typename Traits<T>::frag f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// fragment of 8 consecutive elements starting at element `e0` (multiple of 8) of row `row`
template <typename T>
__device__ __forceinline__ typename Traits<T>::frag lds_frag_row(const char* tile, int row, int e0) {
    const int byte = e0 * (int)sizeof(T);
    const int pan = byte >> 7;
    const int kb = (byte & 127) / (8 * (int)sizeof(T));
    return lds_frag<T>(tile + pan * (64 * 128), row, kb);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void flash_attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ vT, T* __restrict__ ctx,
                                                            int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad) {
    typedef typename Traits<T>::frag frag_t;
    constexpr int NPAN = FaCfg<T>::NPAN;
    constexpr int TILE = FaCfg<T>::TILE_BYTES;
    __shared__ __attribute__((aligned(1024))) char lds[4 * TILE];   // K0 K1 V0 V1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;

    // XCD-aware mapping: all query blocks of one (image, head) pair run on one XCD (its K/V stay in that L2)
    const int nq = n_pad / FA_QROWS;
    const int pairs = B * H;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int pair = (j / nq) * 8 + xcd;
    const int qb = j % nq;
    if (pair >= pairs) return;
    const int b = pair / H, h = pair % H;

    const T* qbase = q + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64;
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * qk_batch_stride + ((int64_t)h * n_pad) * 64);
    const char* vbase = reinterpret_cast<const char*>(vT + ((int64_t)pair * 64) * n_pad);
    const int64_t k_ld = 64 * (int64_t)sizeof(T);
    const int64_t v_ld = (int64_t)n_pad * sizeof(T);

    // Q fragments: qf[qt][ks] = Q[row q0 + qt*16 + l15][d = ks*32 + lg*8 .. +7]
    const int q0 = qb * FA_QROWS + wave * 32;
    frag_t qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[qt][ks] = *reinterpret_cast<const frag_t*>(qbase + (int64_t)(q0 + qt * 16 + l15) * 64 + ks * 32 + lg * 8);

    auto stage = [&](int t, int buf) {
        char* sk = lds + buf * TILE;
        char* sv = lds + (2 + buf) * TILE;
        const int key0 = t * FA_KEYS;
#pragma unroll
        for (int p = 0; p < NPAN; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row8 = (wave * 2 + i) * 8;
                glds_rows8(sk + p * (64 * 128) + row8 * 128, kbase + (int64_t)key0 * k_ld + p * 128, k_ld, row8, lane);
                glds_rows8(sv + p * (64 * 128) + row8 * 128, vbase + (int64_t)key0 * sizeof(T) + p * 128, v_ld, row8, lane);
            }
        }
    };

    f32x4 oacc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) oacc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrow[2] = {-INFINITY, -INFINITY};   // running max (raw score units)
    float lrow[2] = {0.f, 0.f};               // lane-partial running sum

    const int ntiles = (n_valid + FA_KEYS - 1) / FA_KEYS;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) stage(t + 1, buf ^ 1);
        const char* sk = lds + buf * TILE;
        const char* sv = lds + (2 + buf) * TILE;

        // ---- S^T = K Q^T ----
        f32x4 sacc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) sacc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const frag_t kf = lds_frag_row<T>(sk, kt * 16 + l15, ks * 32 + lg * 8);
                sacc[0][kt] = mma(kf, qf[0][ks], sacc[0][kt]);
                sacc[1][kt] = mma(kf, qf[1][ks], sacc[1][kt]);
            }
        }
        // ---- mask the ragged last tile (keys >= n_valid) ----
        const int key0 = t * FA_KEYS;
        if (key0 + FA_KEYS > n_valid) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dead = key0 + kt * 16 + lg * 4 + r >= n_valid;
                    if (dead) { sacc[0][kt][r] = -INFINITY; sacc[1][kt][r] = -INFINITY; }
                }
        }
        // ---- online softmax (per query = per lane column) ----
        frag_t pf[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = sacc[qt][0][0];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qt][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(mrow[qt], mx);
            const float alpha = exp2f((mrow[qt] - mnew) * LOG2E);   // exp2f(-inf) = 0 on the first tile
            const float mb = mnew * LOG2E;
            mrow[qt] = mnew;
            float psum = 0.f;
            float pv[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = exp2f(fmaf(sacc[qt][kt][r], LOG2E, -mb));
                    pv[kt][r] = p;
                    psum += p;
                }
            lrow[qt] = lrow[qt] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] *= alpha;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pf[qt][kk][e] = from_f32<T>(pv[2 * kk][e]);
                    pf[qt][kk][4 + e] = from_f32<T>(pv[2 * kk + 1][e]);
                }
        }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const frag_t vf = lds_frag_split<T>(sv, dt * 16 + l15, kk * 32 + lg * 4, kk * 32 + 16 + lg * 4);
                oacc[0][dt] = mma(vf, pf[0][kk], oacc[0][dt]);
                oacc[1][dt] = mma(vf, pf[1][kk], oacc[1][dt]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: O = O^T / l, ctx[(b*n_pad + q)][h*64 + d] ----
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = lrow[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int qrow = q0 + qt * 16 + l15;
        T* o = ctx + ((int64_t)b * n_pad + qrow) * (H * 64) + h * 64 + lg * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 v = oacc[qt][dt] * inv;
            *reinterpret_cast<typename Traits<T>::vec4*>(o + dt * 16) = pack4<T>(v[0], v[1], v[2], v[3]);
        }
    }
}

hipError_t launch_flash_attn(int dtype, const void* q, const void* k, const void* vT, void* ctx,
                             int64_t qk_batch_stride, int B, int H, int n_valid, int n_pad, hipStream_t s) {
    if (n_pad % FA_QROWS || n_valid <= 0 || n_valid > n_pad || B <= 0 || H <= 0) return hipErrorInvalidValue;
    const int nq = n_pad / FA_QROWS;
    const int pairs = B * H;
    dim3 grid(((pairs + 7) / 8) * 8 * nq), block(256);
    switch (dtype) {
        case DT_F32:
            hipLaunchKernelGGL(flash_attn_kernel<float>, grid, block, 0, s, (const float*)q, (const float*)k,
                               (const float*)vT, (float*)ctx, qk_batch_stride, B, H, n_valid, n_pad);
            break;
        case DT_BF16:
            hipLaunchKernelGGL(flash_attn_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)q, (const bf16_t*)k,
                               (const bf16_t*)vT, (bf16_t*)ctx, qk_batch_stride, B, H, n_valid, n_pad);
            break;
        case DT_F16:
            hipLaunchKernelGGL(flash_attn_kernel<f16_t>, grid, block, 0, s, (const f16_t*)q, (const f16_t*)k,
                               (const f16_t*)vT, (f16_t*)ctx, qk_batch_stride, B, H, n_valid, n_pad);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// MPNet attention (short sequences, run once per prompt set): one thread per (prompt, head, query).
// scores = q.k/8 (1/8 folded into the packed q weights) + bias[h][i][j] + mask_add[t][j];
// mask_add = 0 for attended keys, -FLT_MAX otherwise (HF additive mask) -> fully masked rows become
// uniform exactly like the reference.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void text_attn_kernel(const T* __restrict__ qkv, const float* __restrict__ bias,
                                 const int64_t* __restrict__ mask, T* __restrict__ ctx, int Tn, int L, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y, t = blockIdx.z;
    if (i >= L) return;
    const int D3 = 3 * H * 64, D = H * 64;
    const T* qp = qkv + ((int64_t)t * L + i) * D3 + h * 64;
    float qv[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) qv[d] = to_f32(qp[d]);
    float o[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float* brow = bias + ((int64_t)h * L + i) * L;
    for (int jj = 0; jj < L; ++jj) {
        const T* kp = qkv + ((int64_t)t * L + jj) * D3 + D + h * 64;
        const T* vp = kp + D;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) s = fmaf(qv[d], to_f32(kp[d]), s);
        s += brow[jj];
        s += (mask[(int64_t)t * L + jj] != 0) ? 0.f : -3.4028234663852886e38f;
        const float mn = fmaxf(m, s);
        const float alpha = expf(m - mn);
        const float p = expf(s - mn);
        l = l * alpha + p;
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = fmaf(p, to_f32(vp[d]), o[d] * alpha);
        m = mn;
    }
    const float inv = 1.f / l;
    T* op = ctx + ((int64_t)t * L + i) * D + h * 64;
#pragma unroll
    for (int d = 0; d < 64; ++d) op[d] = from_f32<T>(o[d] * inv);
}

hipError_t launch_text_attn(int dtype, const void* qkv, const float* rel_bias, const int* /*bucket_tbl*/,
                            const int64_t* attn_mask, void* ctx, int T, int L, int H, int /*num_buckets*/,
                            hipStream_t s) {
    if (T <= 0 || L <= 0) return hipErrorInvalidValue;
    dim3 block(64), grid((L + 63) / 64, H, T);
    switch (dtype) {
        case DT_F32:
            hipLaunchKernelGGL(text_attn_kernel<float>, grid, block, 0, s, (const float*)qkv, rel_bias, attn_mask,
                               (float*)ctx, T, L, H);
            break;
        case DT_BF16:
            hipLaunchKernelGGL(text_attn_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)qkv, rel_bias, attn_mask,
                               (bf16_t*)ctx, T, L, H);
            break;
        case DT_F16:
            hipLaunchKernelGGL(text_attn_kernel<f16_t>, grid, block, 0, s, (const f16_t*)qkv, rel_bias, attn_mask,
                               (f16_t*)ctx, T, L, H);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rz
