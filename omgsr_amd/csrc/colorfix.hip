// SURVEY.md §8(f) row f1 — the reference driver's post-process on the GPU: uint8 conversion + AdaIN / wavelet colour
// fix (infer/infer_omgsr_s.py:96-103, infer/wavelet_color_fix.py:12-125). The reference does this on the CPU through
// PIL for every image; at hundreds of images/s it is the end-to-end bottleneck. HBM-bound byte/float work:
// no MFMA, coalesced 16-byte pixel loads, wavefront reductions, integer-exact statistics.
//
//   target  = ToPILImage(clip(out * 0.5 + 0.5, 0, 1).float())     out in the model dtype: the "+ 0.5" rounds to it
//   source  = the upscaled LQ image the model was fed (uint8 HWC)
//   adain   : per-channel (mean, sqrt(unbiased var + 1e-5)) of target/255 and source/255; (t - mt) / st * ss + ms
//   wavelet : 5-level a-trous decomposition ([1 2 1]^2 / 16, dilation 2^i, replicate border);
//             result = sum_i (t_i - t_{i+1}) + s_5
//   result  -> clamp(0, 1) -> mul(255).byte()  (truncation)
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "timing.hip.h"

namespace {

constexpr int CF_PPB = 4096;          // pixels per block of the statistics pass

template <typename T>
OMGSR_DEVINL void target_u8(const T* __restrict__ px, unsigned (&c)[3]) {
    // three channels of one NHWC pixel -> the uint8 values ToPILImage would produce
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float y = (float)(T)((float)px[e] * 0.5f + 0.5f);            // x * 0.5 is exact; the add rounds to T (torch op-by-op)
        const float z = fminf(fmaxf(y, 0.0f), 1.0f);
        c[e] = (unsigned)(z * 255.0f);                                       // truncation
    }
}

// integer-exact per-(image, channel) sums of the uint8 target and source: [N][nblk][12] u64
// order: t_sum[3], t_sq[3], s_sum[3], s_sq[3]
template <typename T>
__global__ __launch_bounds__(256) void cf_stats_kernel(const T* __restrict__ sr, int sr_ld, const unsigned char* __restrict__ src,
                                                        unsigned long long* __restrict__ partial, int64_t HW) {
    const int n = blockIdx.y, t = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * CF_PPB;
    int64_t p1 = p0 + CF_PPB; if (p1 > HW) p1 = HW;
    unsigned acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0u;                               // 4096 px * 255^2 / 256 threads fits 32 bits
    for (int64_t p = p0 + t; p < p1; p += 256) {
        unsigned c[3];
        target_u8<T>(sr + ((int64_t)n * HW + p) * sr_ld, c);
        const unsigned char* s = src + ((int64_t)n * HW + p) * 3;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const unsigned sv = s[e];
            acc[e] += c[e]; acc[3 + e] += c[e] * c[e];
            acc[6 + e] += sv; acc[9 + e] += sv * sv;
        }
    }
    __shared__ unsigned long long red[4][12];
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        unsigned long long v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (t < 12) partial[((int64_t)n * gridDim.x + blockIdx.x) * 12 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// stats [N][12] f32: per channel (t_mean, t_std, s_mean, s_std) as calc_mean_std computes them on x / 255
__global__ void cf_finalize_kernel(const unsigned long long* __restrict__ partial, float* __restrict__ stats, int nblk, int64_t HW) {
    const int n = blockIdx.x, t = threadIdx.x;          // 64 threads; thread i < 12 owns one sum
    __shared__ double tot[12];
    if (t < 12) {
        unsigned long long s = 0;
        for (int b = 0; b < nblk; ++b) s += partial[((int64_t)n * nblk + b) * 12 + t];
        tot[t] = (double)s;
    }
    __syncthreads();
    if (t < 6) {
        const int ch = t % 3, which = t / 3;            // which: 0 target, 1 source
        const double cnt = (double)HW;
        const double sum = tot[which * 6 + ch], sq = tot[which * 6 + 3 + ch];
        const double mean = sum / cnt / 255.0;
        double var = (sq - sum * sum / cnt) / (cnt - 1.0) / (255.0 * 255.0);      // unbiased, as torch.var
        if (var < 0.0) var = 0.0;
        stats[n * 12 + which * 6 + ch] = (float)mean;
        stats[n * 12 + which * 6 + 3 + ch] = (float)sqrt((double)((float)var + 1e-5f));
    }
}

// method 0: plain uint8 conversion; method 1: AdaIN with `stats`
template <typename T>
__global__ __launch_bounds__(256) void cf_apply_kernel(const T* __restrict__ sr, int sr_ld, const float* __restrict__ stats,
                                                        unsigned char* __restrict__ out, int64_t HW, int method) {
    const int n = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    unsigned c[3];
    target_u8<T>(sr + ((int64_t)n * HW + p) * sr_ld, c);
    unsigned char* o = out + ((int64_t)n * HW + p) * 3;
    if (method == 0) {
        o[0] = (unsigned char)c[0]; o[1] = (unsigned char)c[1]; o[2] = (unsigned char)c[2];
        return;
    }
    const float* st = stats + n * 12;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        // (content - mean_c) / std_c * std_s + mean_s, op by op in fp32 (no FMA contraction: the byte truncation sees it)
        const float x = __fdiv_rn((float)c[e], 255.0f);
        float v = __fsub_rn(x, st[e]);
        v = __fdiv_rn(v, st[3 + e]);
        v = __fmul_rn(v, st[9 + e]);
        v = __fadd_rn(v, st[6 + e]);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        o[e] = (unsigned char)(unsigned)__fmul_rn(v, 255.0f);
    }
}

// ---- wavelet ----------------------------------------------------------------------------------
// planes: fp32 [2 (target, source)][N][3][H][W]
template <typename T>
__global__ __launch_bounds__(256) void cf_planes_kernel(const T* __restrict__ sr, int sr_ld, const unsigned char* __restrict__ src,
                                                         float* __restrict__ planes, int64_t HW, int N) {
    const int n = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    unsigned c[3];
    target_u8<T>(sr + ((int64_t)n * HW + p) * sr_ld, c);
    const unsigned char* s = src + ((int64_t)n * HW + p) * 3;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        planes[((int64_t)(0 * N + n) * 3 + e) * HW + p] = __fdiv_rn((float)c[e], 255.0f);
        planes[((int64_t)(1 * N + n) * 3 + e) * HW + p] = __fdiv_rn((float)s[e], 255.0f);
    }
}

// one a-trous level on every plane: low = blur(in, radius); target planes also accumulate high += in - low
__global__ __launch_bounds__(256) void cf_level_kernel(const float* __restrict__ in, float* __restrict__ low, float* __restrict__ high,
                                                        int H, int W, int radius, int n_target_planes, int first) {
    const int plane = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
    const float* img = in + (int64_t)plane * HW;
    const int ym = max(y - radius, 0), yp = min(y + radius, H - 1);
    const int xm = max(x - radius, 0), xp = min(x + radius, W - 1);
    const float* r0 = img + (int64_t)ym * W;
    const float* r1 = img + (int64_t)y * W;
    const float* r2 = img + (int64_t)yp * W;
    float a = 0.0625f * r0[xm];
    a = __fadd_rn(a, 0.125f * r0[x]);
    a = __fadd_rn(a, 0.0625f * r0[xp]);
    a = __fadd_rn(a, 0.125f * r1[xm]);
    a = __fadd_rn(a, 0.25f * r1[x]);
    a = __fadd_rn(a, 0.125f * r1[xp]);
    a = __fadd_rn(a, 0.0625f * r2[xm]);
    a = __fadd_rn(a, 0.125f * r2[x]);
    a = __fadd_rn(a, 0.0625f * r2[xp]);
    low[(int64_t)plane * HW + p] = a;
    if (plane < n_target_planes) {
        const float d = __fsub_rn(r1[x], a);
        float* h = high + (int64_t)plane * HW + p;
        *h = first ? d : __fadd_rn(*h, d);
    }
}

__global__ __launch_bounds__(256) void cf_wavelet_out_kernel(const float* __restrict__ high, const float* __restrict__ low_src,
                                                              unsigned char* __restrict__ out, int64_t HW) {
    const int n = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        float v = __fadd_rn(high[((int64_t)n * 3 + e) * HW + p], low_src[((int64_t)n * 3 + e) * HW + p]);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        out[((int64_t)n * HW + p) * 3 + e] = (unsigned char)(unsigned)__fmul_rn(v, 255.0f);
    }
}

// uint8 HWC image -> model input NHWC (8 channels, 5 zero): F.to_tensor(img).to(dtype) * 2 - 1 (infer/infer_omgsr_s.py:92),
// rounded to the model dtype after the conversion AND after the affine, as torch does op by op
template <typename T, int OEL>      // OEL 1: T = float, fp32 stream tensor out (accurate tier)
__global__ __launch_bounds__(256) void cf_input_kernel(const unsigned char* __restrict__ img, void* __restrict__ out, int64_t total) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const unsigned char* s = img + p * 3;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.0f;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float x = (float)(T)__fdiv_rn((float)s[e], 255.0f);
        f[e] = __fsub_rn(x * 2.0f, 1.0f);
    }
    if constexpr (OEL == 1) store8<bf16_t, 1>(out, p * 8, 0, f);
    else store8<T, 0>(out, p * 8, 0, f);
}

int64_t stats_blocks(int64_t HW) { return (HW + CF_PPB - 1) / CF_PPB; }

}  // namespace

extern "C" int omgsr_image_to_model_input(const uint8_t* img_hwc3, void* out_nhwc8, int32_t N, int32_t H, int32_t W, int32_t out_el,
                                          void* stream) {
    if (!img_hwc3 || !out_nhwc8 || N <= 0 || H <= 0 || W <= 0 || (out_el != OMGSR_EL_16 && out_el != OMGSR_EL_F32)) return OMGSR_E_BADARG;
    const int64_t total = (int64_t)N * H * W;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (out_el == OMGSR_EL_F32) hipLaunchKernelGGL((cf_input_kernel<float, 1>), grid, dim3(256), 0, (hipStream_t)stream, img_hwc3, out_nhwc8, total);
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((cf_input_kernel<T, 0>), grid, dim3(256), 0, (hipStream_t)stream, img_hwc3, out_nhwc8, total));
    return (int)hipGetLastError();
}

extern "C" int64_t omgsr_colorfix_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t method) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t HW = (int64_t)H * W;
    if (method == OMGSR_COLORFIX_ADAIN) return (N * stats_blocks(HW) * 12) * 8 + (int64_t)N * 12 * 4;
    if (method == OMGSR_COLORFIX_WAVELET) return (int64_t)(2 + 2 + 1) * N * 3 * HW * 4;    // ping, pong (both images), high (target)
    return 0;
}

extern "C" int omgsr_colorfix(const void* sr_nhwc, int32_t sr_ld, const uint8_t* src_hwc3, uint8_t* out_hwc3, void* workspace,
                              int32_t N, int32_t H, int32_t W, int32_t method, int32_t sr_el, void* stream) {
    if (!sr_nhwc || !out_hwc3 || N <= 0 || H <= 0 || W <= 0 || sr_ld < 3) return OMGSR_E_BADARG;
    if (sr_el != OMGSR_EL_16 && sr_el != OMGSR_EL_F32) return OMGSR_E_BADARG;
    // sr_el = OMGSR_EL_F32: the model ran in the accurate tier (--weight_dtype fp32): the "* 0.5 + 0.5" then rounds to fp32
#define OMGSR_CF_DISPATCH(...)                                              \
    do {                                                                    \
        if (sr_el == OMGSR_EL_F32) { using T = float; __VA_ARGS__; }        \
        else OMGSR_DISPATCH_T(__VA_ARGS__);                                 \
    } while (0)
    if (method != OMGSR_COLORFIX_NONE && method != OMGSR_COLORFIX_ADAIN && method != OMGSR_COLORFIX_WAVELET) return OMGSR_E_BADARG;
    if (method != OMGSR_COLORFIX_NONE && (!src_hwc3 || !workspace)) return OMGSR_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    if (HW < 2 || HW >= (1ll << 31)) return OMGSR_E_SHAPE;
    const dim3 px_grid((unsigned)((HW + 255) / 256), N);
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, (double)N * HW * (2.0 * sr_ld * (method == OMGSR_COLORFIX_ADAIN ? 2 : 1) + 6.0), st);
    if (method == OMGSR_COLORFIX_WAVELET) {
        float* ping = (float*)workspace;
        float* pong = ping + (int64_t)2 * N * 3 * HW;
        float* high = pong + (int64_t)2 * N * 3 * HW;
        OMGSR_CF_DISPATCH(hipLaunchKernelGGL(cf_planes_kernel<T>, px_grid, dim3(256), 0, st, (const T*)sr_nhwc, sr_ld, src_hwc3, ping, HW, N));
        const dim3 lvl_grid((unsigned)((HW + 255) / 256), 2 * N * 3);
        for (int i = 0; i < 5; ++i) {
            hipLaunchKernelGGL(cf_level_kernel, lvl_grid, dim3(256), 0, st, ping, pong, high, H, W, 1 << i, N * 3, i == 0 ? 1 : 0);
            float* tmp = ping; ping = pong; pong = tmp;
        }
        hipLaunchKernelGGL(cf_wavelet_out_kernel, px_grid, dim3(256), 0, st, high, ping + (int64_t)N * 3 * HW, out_hwc3, HW);
        return (int)hipGetLastError();
    }
    float* stats = nullptr;
    if (method == OMGSR_COLORFIX_ADAIN) {
        const int nblk = (int)stats_blocks(HW);
        unsigned long long* partial = (unsigned long long*)workspace;
        stats = (float*)(partial + (int64_t)N * nblk * 12);
        OMGSR_CF_DISPATCH(hipLaunchKernelGGL(cf_stats_kernel<T>, dim3(nblk, N), dim3(256), 0, st, (const T*)sr_nhwc, sr_ld, src_hwc3, partial, HW));
        hipLaunchKernelGGL(cf_finalize_kernel, dim3(N), dim3(64), 0, st, partial, stats, nblk, HW);
    }
    OMGSR_CF_DISPATCH(hipLaunchKernelGGL(cf_apply_kernel<T>, px_grid, dim3(256), 0, st, (const T*)sr_nhwc, sr_ld, stats, out_hwc3, HW, method));
#undef OMGSR_CF_DISPATCH
    return (int)hipGetLastError();
}
