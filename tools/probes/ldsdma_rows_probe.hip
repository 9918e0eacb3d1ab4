// Probe (build: hipcc --offload-arch=gfx950 -O2 ldsdma_rows_probe.hip -o ldsdma_rows_probe): how fast does global_load_lds_dwordx4
// (LDS-DMA) stream a row-major operand [M][row_bytes] when a K-step takes (a) 64 bytes of every row (a wave instruction = 16 rows x 64 B:
// sixteen HALF 128-byte lines - what igemm_dma_kernel / igemm_gmx_kernel do with BK = 32) or (b) 128 bytes (8 rows x 128 B: eight FULL
// lines - igemm_p8_kernel's BK = 64)? Same bytes per workgroup (a 256-row tile, all of K), two workgroups per CU, nothing but the DMA
// stream and the counted waits. Usage: ldsdma_rows_probe [row_bytes=640] [M=147456] [passes over the tile set=3]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// MODE 0: 16 rows x 64 B per piece; MODE 1: 8 rows x 128 B per piece. A 256-row tile: 16 KB per 64-byte step (4 pieces per wave) in
// mode 0, 32 KB per 128-byte step (8 pieces per wave) in mode 1. Ring of 48 KB either way.
template <int MODE>
__global__ __launch_bounds__(256, 2) void stream(const unsigned char* __restrict__ a, int M, int row_bytes, int reps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    const unsigned lds_base = (unsigned)(size_t)(lds_byte_t*)lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntile = M / 256;
    for (int r = 0; r < reps; ++r) {
        for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
            const int m0 = tile * 256;
            if (MODE == 0) {
                const unsigned char* p[4];
                for (int i = 0; i < 4; ++i) p[i] = a + (size_t)(m0 + 16 * (wave * 4 + i) + (lane >> 2)) * row_bytes + (lane & 3) * 16;
                const int steps = row_bytes / 64;
                for (int s = 0; s < steps; ++s) {
                    const unsigned dst = lds_base + (s % 3) * 16384 + wave * 4096;
                    for (int i = 0; i < 4; ++i) { glds16(p[i], __builtin_amdgcn_readfirstlane(dst + i * 1024)); p[i] += 64; }
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
            } else {
                const unsigned char* p[8];
                for (int i = 0; i < 8; ++i) p[i] = a + (size_t)(m0 + 8 * (wave * 8 + i) + (lane >> 3)) * row_bytes + (lane & 7) * 16;
                const int steps = row_bytes / 128;
                for (int s = 0; s < steps; ++s) {
                    const unsigned dst = lds_base + (s & 1) * 32768 + wave * 8192;
                    for (int i = 0; i < 8; ++i) { glds16(p[i], __builtin_amdgcn_readfirstlane(dst + i * 1024)); p[i] += 128; }
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = lds[0];
}

int main(int argc, char** argv) {
    const int row_bytes = argc > 1 ? atoi(argv[1]) : 640, M = argc > 2 ? atoi(argv[2]) : 147456, reps = argc > 3 ? atoi(argv[3]) : 3;
    unsigned char* a; unsigned* sink;
    const size_t bytes = (size_t)M * row_bytes;
    hipMalloc(&a, bytes); hipMemset(a, 1, bytes); hipMalloc(&sink, 4096 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 1024}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e30f;
            for (int it = 0; it < 5; ++it) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(grid), dim3(256), 65536, 0, a, M, row_bytes, reps, sink);
                else hipLaunchKernelGGL(stream<1>, dim3(grid), dim3(256), 65536, 0, a, M, row_bytes, reps, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("row_bytes %d M %d (%.0f MB) x %d passes, grid %d, %s: %.1f us per pass, %.2f TB/s\n", row_bytes, M, bytes / 1e6, reps, grid,
                   mode == 0 ? "16 rows x 64 B pieces " : " 8 rows x 128 B pieces", best * 1e3 / reps, bytes * (double)reps / (best * 1e-3) / 1e12);
        }
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
