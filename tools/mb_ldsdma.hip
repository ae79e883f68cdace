// Micro-benchmark (measurement tool, not product): how fast can one workgroup per CU stream operand panels
// L2 -> LDS with global_load_lds_dwordx4, as a function of ring depth (stages kept in flight across the barrier).
// Mimics gemm_kernel_v3's staging: per K-step a 256-row x 128-B A panel (rows `lda` bytes apart, distinct per
// workgroup) and a 256-row x 128-B W panel (shared by all workgroups).   hipcc --offload-arch=gfx950 -O3 ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEPTH, int STAGE_KB>   // STAGE_KB: 64 (A+W), 32 (A only)
__global__ __launch_bounds__(512, 2) void stream_kernel(const char* A, const char* W, int64_t lda, int64_t ldw, int nk, int ntile_iters,
                                                         int tiles_m, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER_WAVE = STAGE_KB / 8;         // 1-KB glds per wave per stage
    auto stage = [&](const char* ga, const char* gw, int kt, int buf) {
        char* s = lds + buf * (STAGE_KB * 1024);
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int piece = wave * PER_WAVE + i;            // 1-KB piece = 8 rows x 128 B
            const bool isw = (STAGE_KB == 64) && piece >= 32;
            const int row8 = (isw ? piece - 32 : piece) * 8;
            const int r = row8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            const char* src = (isw ? gw + (int64_t)r * ldw : ga + (int64_t)r * lda) + (int64_t)kt * 128 + (c << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(s + piece * 1024), 16, 0, 0);
        }
    };
    float acc = 0.f;
    for (int it = 0; it < ntile_iters; ++it) {
        const int tm = (blockIdx.x + it * gridDim.x) % tiles_m;
        const char* ga = A + (int64_t)tm * 256 * (lda < 1024 ? 1024 : lda);
        const char* gw = W;
        // prologue: DEPTH stages in flight
        for (int d = 0; d < DEPTH && d < nk; ++d) stage(ga, gw, d, d % (DEPTH + 1));
        for (int kt = 0; kt < nk; ++kt) {
            // retire stage kt: allow (DEPTH-1) newer stages in flight
            const int newer = (nk - 1 - kt) < (DEPTH - 1) ? (nk - 1 - kt) : (DEPTH - 1);
            if (newer == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (newer == 1) { if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else { if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            __builtin_amdgcn_s_barrier();
            // "consume": one LDS read so the data dependency is real
            acc += *reinterpret_cast<const float*>(lds + (kt % (DEPTH + 1)) * (STAGE_KB * 1024) + tid * 4);
            __builtin_amdgcn_s_barrier();
            if (kt + DEPTH < nk) stage(ga, gw, kt + DEPTH, (kt + DEPTH) % (DEPTH + 1));
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int DEPTH, int STAGE_KB>
static void run(const char* name, const char* A, const char* W, int64_t lda, int64_t ldw, int nk, int iters, int tiles_m, float* sink, int grid = 256) {
    const size_t shm = (size_t)(DEPTH + 1) * STAGE_KB * 1024;
    if (shm > 160 * 1024) { printf("%s: skip (LDS %zu)\n", name, shm); return; }
    hipFuncSetAttribute((const void*)stream_kernel<DEPTH, STAGE_KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((stream_kernel<DEPTH, STAGE_KB>), dim3(grid), dim3(512), shm, 0, A, W, lda, ldw, nk, iters, tiles_m, sink);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * iters * nk * STAGE_KB * 1024.0;
    printf("%-28s grid=%d depth=%d stage=%dKB nk=%d: %.3f ms  %.2f TB/s  %.1f GB/s/CU  %.2f us/K-step\n", name, grid, DEPTH, STAGE_KB, nk, ms, bytes / ms / 1e9,
           bytes / ms / 1e6 / grid, ms * 1e3 / (iters * nk));
}

// Same traffic by ordinary vector loads: each wave fetches its 1-KB pieces with global_load_dwordx4 into VGPRs
// (WRITE: and stores them to LDS with ds_write_b128), PER_WAVE pieces per K step, next step's loads issued before
// the current step's registers are consumed (register double buffer).
template <int STAGE_KB, bool WRITE>
__global__ __launch_bounds__(512, 2) void plain_kernel(const char* A, const char* W, int64_t lda, int64_t ldw, int nk, int ntile_iters,
                                                        int tiles_m, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER_WAVE = STAGE_KB / 8;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 cur[PER_WAVE], nxt[PER_WAVE];
    auto fetch = [&](f4 (&dst)[PER_WAVE], const char* ga, const char* gw, int kt) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int piece = wave * PER_WAVE + i;
            const bool isw = (STAGE_KB == 64) && piece >= 32;
            const int row8 = (isw ? piece - 32 : piece) * 8;
            const int r = row8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            const char* src = (isw ? gw + (int64_t)r * ldw : ga + (int64_t)r * lda) + (int64_t)kt * 128 + (c << 4);
            dst[i] = *reinterpret_cast<const f4*>(src);
        }
    };
    float acc = 0.f;
    for (int it = 0; it < ntile_iters; ++it) {
        const int tm = (blockIdx.x + it * gridDim.x) % tiles_m;
        const char* ga = A + (int64_t)tm * 256 * (lda < 1024 ? 1024 : lda);
        fetch(cur, ga, W, 0);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) fetch(nxt, ga, W, kt + 1);
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) {
                if (WRITE) *reinterpret_cast<f4*>(lds + (kt & 1) * (STAGE_KB * 1024) + (wave * PER_WAVE + i) * 1024 + lane * 16) = cur[i];
                else acc += cur[i][0] + cur[i][3];
            }
            if (WRITE) { __syncthreads(); acc += *reinterpret_cast<const float*>(lds + (kt & 1) * (STAGE_KB * 1024) + tid * 4); }
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) cur[i] = nxt[i];
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int STAGE_KB, bool WRITE>
static void run_plain(const char* name, const char* A, const char* W, int64_t lda, int64_t ldw, int nk, int iters, int tiles_m, float* sink, int grid = 256) {
    const size_t shm = 2 * STAGE_KB * 1024;
    hipFuncSetAttribute((const void*)plain_kernel<STAGE_KB, WRITE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((plain_kernel<STAGE_KB, WRITE>), dim3(grid), dim3(512), shm, 0, A, W, lda, ldw, nk, iters, tiles_m, sink);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * iters * nk * STAGE_KB * 1024.0;
    printf("%-28s grid=%d plain loads%s stage=%dKB nk=%d: %.3f ms  %.2f TB/s  %.1f GB/s/CU  %.2f us/K-step\n", name, grid, WRITE ? "+ds_write" : "", STAGE_KB, nk, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / grid, ms * 1e3 / (iters * nk));
}

int main() {
    const int K = 768, M = 43008, N = 3072;
    char *A, *W; float* sink;
    hipMalloc(&A, (size_t)M * 3072 * 2); hipMalloc(&W, (size_t)N * 3072 * 2); hipMalloc(&sink, 4);
    hipMemset(A, 1, (size_t)M * 3072 * 2); hipMemset(W, 1, (size_t)N * 3072 * 2);
    for (int pass = 0; pass < 2; ++pass) {
        const int k = pass == 0 ? K : 3072;
        const int nk = k * 2 / 128, iters = pass == 0 ? 8 : 2;
        printf("--- K=%d (row stride %d B) ---\n", k, k * 2);
        run<1, 64>("A+W", A, W, k * 2, k * 2, nk, iters, M / 256, sink);
        run<1, 32>("A only", A, W, k * 2, k * 2, nk, iters, M / 256, sink);
        run<2, 32>("A only", A, W, k * 2, k * 2, nk, iters, M / 256, sink);
        run<3, 32>("A only", A, W, k * 2, k * 2, nk, iters, M / 256, sink);
        run<4, 32>("A only", A, W, k * 2, k * 2, nk, iters, M / 256, sink);
        run<1, 32>("A only, 21 m-tiles (L2 hot)", A, W, k * 2, k * 2, nk, iters, 21, sink);
        run<3, 32>("A only, 21 m-tiles (L2 hot)", A, W, k * 2, k * 2, nk, iters, 21, sink);
    }
    printf("--- who owns the limit: fewer active CUs, and ordinary vector loads instead of LDS-DMA (K=768) ---\n");
    for (int grid : {256, 128, 64, 32}) run<1, 64>("A+W", A, W, 1536, 1536, 12, 8, M / 256, sink, grid);
    for (int grid : {256, 64}) run<1, 64>("A+W, 4 m-tiles (L2 resident)", A, W, 1536, 1536, 12, 8, 4, sink, grid);
    for (int grid : {256, 64}) run<3, 32>("A only, 4 m-tiles (L2 res.)", A, W, 1536, 1536, 12, 8, 4, sink, grid);
    for (int grid : {256, 64}) run_plain<64, false>("A+W, 4 m-tiles (L2 resident)", A, W, 1536, 1536, 12, 8, 4, sink, grid);
    for (int grid : {256, 64}) run_plain<64, true>("A+W, 4 m-tiles (L2 resident)", A, W, 1536, 1536, 12, 8, 4, sink, grid);
    for (int grid : {256, 64}) {
        run_plain<64, false>("A+W", A, W, 1536, 1536, 12, 8, M / 256, sink, grid);
        run_plain<64, true>("A+W", A, W, 1536, 1536, 12, 8, M / 256, sink, grid);
        run_plain<32, false>("A only", A, W, 1536, 1536, 12, 8, M / 256, sink, grid);
    }
    printf("--- blocked layout: each stage = one contiguous 32/64 KB block (lda = 128 B, stage stride via kt*128 -> use big lda trick) ---\n");
    // emulate [tile][kt][256][64]: row stride 128 B, K-step stride handled by passing lda=128 and nk=1 repeated
    run<1, 32>("A blocked (contig 32 KB)", A, W, 128, 128, 1, 96, 900, sink);
    run<1, 64>("A+W blocked (contig)", A, W, 128, 128, 1, 96, 900, sink);
    return 0;
}
