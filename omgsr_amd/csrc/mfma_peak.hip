// MFMA micro-benchmark: the dense 16-bit matrix-core rate this chip sustains with nothing else in the way
// (v_mfma_f32_32x32x16_{bf16,f16}, four independent accumulators per wave, two waves per SIMD, non-trivial operand
// bits so the clock sits where a real kernel's does). bench.py records it next to every roofline fraction
// ("the peak actually used and a measured MFMA micro-benchmark peak").
#include "common.hip.h"
#include "../../include/omgsr_hip.h"

namespace {
template <typename T>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* __restrict__ sink, const int iters) {
    const int lane = threadIdx.x & 63;
    x8_t<T> a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (T)(0.37f * (float)((lane * 7 + j * 3) % 13 - 6));
        b[j] = (T)(0.21f * (float)((lane * 5 + j * 11) % 17 - 8));
    }
    f32x16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma32(a, b, acc[i]);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123456.789f) sink[blockIdx.x * 256 + threadIdx.x] = s;      // keeps the chain alive, never true
}
}  // namespace

// Runs the micro-benchmark on the current device and returns the measured TFLOP/s through *tflops (HOST pointer).
// Synchronises the stream: a benchmarking utility, not part of the data path.
extern "C" int omgsr_mfma_peak(int32_t iters, float* tflops, void* stream) {
    if (!tflops || iters <= 0) return OMGSR_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return OMGSR_E_ARCH;
    const int blocks = prop.multiProcessorCount * 2;        // 2 x 4 waves per CU = two waves per SIMD
    float* sink = nullptr;
    if (hipMalloc(&sink, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) return OMGSR_E_BADARG;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 0.0f;
    for (int rep = 0; rep < 4; ++rep) {                     // rep 0 warms up (clock ramp, code fetch)
        (void)hipEventRecord(e0, st);
        OMGSR_DISPATCH_T(hipLaunchKernelGGL(mfma_peak_kernel<T>, dim3(blocks), dim3(256), 0, st, sink, iters));
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipFree(sink); return (int)hipGetLastError(); }
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4.0 * (double)iters * 16.0 * 2.0 * 32.0 * 32.0 * 16.0;
        const float tf = (float)(flops / (ms * 1e-3) / 1e12);
        if (rep > 0 && tf > best) best = tf;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    *tflops = best;
    return 0;
}
