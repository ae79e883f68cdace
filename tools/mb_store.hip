// Micro-benchmark: how fast can ONE CU (one 512-thread workgroup) write / read-modify-write an output tile, alone on the
// chip and with every other CU doing the same?  Decides whether the GEMM epilogue is bound by the chip (HBM) or by the CU.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_store.hip -o /tmp/mb_store && /tmp/mb_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: bf16 tile store: 256 rows x 512 B, row stride `ld` bytes; each wave instruction = 8 rows x 128 B (16 B / lane)
// mode 1: the same bytes as 2 rows x 512 B per wave instruction
// mode 2: fp32 read-modify-write of a 256 x 1024 B tile (16 B / lane, 4 rows x 256 B per instruction), 8 loads batched
template <int MODE>
__global__ __launch_bounds__(512) void k(char* base, long ld, long tile_stride, int tiles, int active_mod, int active_rem) {
    if ((int)(blockIdx.x % active_mod) != active_rem && active_mod > 0) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int t = 0; t < tiles; ++t) {
        char* tb = base + ((long)blockIdx.x * tiles + t) * tile_stride;
        if (MODE == 0) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = wave * 32 + (it >> 2) * 8 + (lane >> 3);
                const int col = (it & 3) * 128 + (lane & 7) * 16;
                *reinterpret_cast<f32x4*>(tb + row * ld + col) = (f32x4){1.f, 2.f, 3.f, (float)it};
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int q = it * 512 + tid;
                const int row = q >> 5, c16 = q & 31;
                *reinterpret_cast<f32x4*>(tb + row * ld + c16 * 16) = (f32x4){1.f, 2.f, 3.f, (float)it};
            }
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = (b * 8 + i) * 512 + tid;          // 16-byte chunk index in the 256 x 64-chunk tile
                    const int row = q >> 6, c = q & 63;
                    v[i] = *reinterpret_cast<const f32x4*>(tb + row * ld + c * 16);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = (b * 8 + i) * 512 + tid;
                    const int row = q >> 6, c = q & 63;
                    *reinterpret_cast<f32x4*>(tb + row * ld + c * 16) = v[i] * 1.5f + 1.0f;
                }
            }
        }
    }
}

template <int MODE> float run(char* buf, long ld, long tile_stride, int tiles, int mod, int rem, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, buf, ld, tile_stride, tiles, mod, rem);
    hipEventRecord(a);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, buf, ld, tile_stride, tiles, mod, rem);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main() {
    const int grid = 256, tiles = 32;
    const long ld16 = 1536, ld32 = 3072;                 // N = 768 columns of bf16 / fp32
    // a tile = 256 rows; consecutive tiles of a workgroup are 256 rows apart
    const long ts16 = 256 * ld16, ts32 = 256 * ld32;
    size_t bytes = (size_t)grid * tiles * ts32 + (1 << 20);
    char* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
    struct { const char* name; int mod, rem; } sets[] = {{"all 256 CUs", 1, 0}, {"1 of 2", 2, 0}, {"1 of 4", 4, 0}, {"1 of 8 (one XCD)", 8, 0}, {"1 of 32", 32, 0}, {"1 of 256", 256, 0}};
    for (auto& s : sets) {
        int active = 0; for (int b = 0; b < grid; ++b) active += (b % s.mod) == s.rem;
        float m0 = run<0>(buf, ld16, ts16, tiles, s.mod, s.rem, grid);
        float m1 = run<1>(buf, ld16, ts16, tiles, s.mod, s.rem, grid);
        float m2 = run<2>(buf, ld32, ts32, tiles, s.mod, s.rem, grid);
        const double kb16 = 128.0, kb32 = 512.0;   // bytes moved per tile: 128 KB written | 256 KB read + 256 KB written
        printf("%-18s active WGs %3d | 16-bit store 8x128B: %6.2f us/tile %6.1f GB/s/CU %5.2f TB/s | 2x512B: %6.2f us/tile %6.1f GB/s/CU | fp32 RMW: %6.2f us/tile %6.1f GB/s/CU %5.2f TB/s\n",
               s.name, active, m0 * 1e3 / tiles, kb16 * 1024 / (m0 * 1e3 / tiles) / 1e3, active * kb16 * 1024 / (m0 * 1e3 / tiles) / 1e6,
               m1 * 1e3 / tiles, kb16 * 1024 / (m1 * 1e3 / tiles) / 1e3,
               m2 * 1e3 / tiles, kb32 * 1024 / (m2 * 1e3 / tiles) / 1e3, active * kb32 * 1024 / (m2 * 1e3 / tiles) / 1e6);
    }
    return 0;
}
