// SURVEY.md §8(f) row f2 — the reference driver's PRE-process on the GPU: Pillow's 8-bit image resampler, bit for bit.
// infer/infer_omgsr_s.py:71-84 resizes every input on the CPU through PIL (bicubic x4, then a LANCZOS snap of width / height
// to multiples of 8) before the model sees it; at tens of images per second per GPU that is the other host-side stage next
// to the colour fix (colorfix.hip). Pillow's algorithm (libImaging/Resample.c, ImagingResampleHorizontal/Vertical_8bpc):
// per output index a window [first, first + count) of input pixels and `count` fixed-point weights (2^22 scale, computed on
// the host in float64 exactly like Pillow's C code: omgsr_amd/preprocess.py); acc = 2^21 + sum(pixel * weight) in int32;
// out = clip8(acc >> 22). HBM-bound byte / integer work: one thread per output pixel, the window of neighbouring threads
// overlaps (horizontal pass) or is the same rows at neighbouring columns (vertical pass), so lines are reused from L1/L2.
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "timing.hip.h"

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;

OMGSR_DEVINL unsigned char clip8(int v) {
    v >>= PRECISION_BITS;                       // arithmetic shift, like Pillow's lookup table index
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// AXIS 1: horizontal (out_size = output width), AXIS 0: vertical (out_size = output height)
template <int AXIS>
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                           const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                           int Hin, int Win, int Hout, int Wout) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (i >= (int64_t)Hout * Wout) return;
    const int oy = (int)(i / Wout), ox = (int)(i - (int64_t)oy * Wout);
    const int o = AXIS == 1 ? ox : oy;
    const int first = bounds[2 * o], count = bounds[2 * o + 1];
    const int* k = kk + (int64_t)o * ksize;
    const unsigned char* s = src + (int64_t)n * Hin * Win * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    if (AXIS == 1) {
        const unsigned char* p = s + ((int64_t)oy * Win + first) * 3;
        for (int x = 0; x < count; ++x) {
            const int w = k[x];
            a0 += (int)p[3 * x] * w; a1 += (int)p[3 * x + 1] * w; a2 += (int)p[3 * x + 2] * w;
        }
    } else {
        const unsigned char* p = s + ((int64_t)first * Win + ox) * 3;
        const int64_t pitch = (int64_t)Win * 3;
        for (int y = 0; y < count; ++y) {
            const int w = k[y];
            a0 += (int)p[y * pitch] * w; a1 += (int)p[y * pitch + 1] * w; a2 += (int)p[y * pitch + 2] * w;
        }
    }
    unsigned char* d = dst + (((int64_t)n * Hout + oy) * Wout + ox) * 3;
    d[0] = clip8(a0); d[1] = clip8(a1); d[2] = clip8(a2);
}
}  // namespace

extern "C" int omgsr_resample_u8(const uint8_t* src, uint8_t* dst, const int32_t* bounds, const int32_t* kk, int32_t ksize,
                                 int32_t N, int32_t Hin, int32_t Win, int32_t out_size, int32_t axis, void* stream) {
    if (!src || !dst || !bounds || !kk || N <= 0 || Hin <= 0 || Win <= 0 || out_size <= 0 || ksize <= 0) return OMGSR_E_BADARG;
    if (axis != 0 && axis != 1) return OMGSR_E_BADARG;
    const int Hout = axis == 0 ? out_size : Hin, Wout = axis == 1 ? out_size : Win;
    const int64_t px = (int64_t)Hout * Wout;
    if (px >= (1ll << 31)) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 3.0 * N * ((double)Hin * Win + (double)px), st);
    const dim3 grid((unsigned)((px + 255) / 256), N);
    if (axis == 1) hipLaunchKernelGGL(resample_u8_kernel<1>, grid, dim3(256), 0, st, src, dst, bounds, kk, ksize, Hin, Win, Hout, Wout);
    else hipLaunchKernelGGL(resample_u8_kernel<0>, grid, dim3(256), 0, st, src, dst, bounds, kk, ksize, Hin, Win, Hout, Wout);
    return (int)hipGetLastError();
}
