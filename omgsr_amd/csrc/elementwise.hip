// Layout changes and latent algebra plus the library's housekeeping entry
// points. Everything here is HBM-bound plumbing around the MFMA kernels.
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "timing.hip.h"
#include <string.h>
#include <type_traits>

namespace omgsr {
TimingState& timing_state() { static TimingState s; return s; }
static int g_compute_dtype = 0;
int compute_dtype() { return g_compute_dtype; }
}

namespace {

template <typename T> OMGSR_DEVINL float round_to(float x) { return (float)(T)x; }

// NCHW -> NHWC(Cpad), 16-bit or fp32 (DEL). One thread per output pixel-channel-group of 8.
template <typename SRC, typename T, int DEL>
__global__ void nchw_to_nhwc_kernel(const SRC* __restrict__ src, void* __restrict__ dst, int N, int C, int64_t HW, int Cpad) {
    const int ng = Cpad >> 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * HW * ng;
    if (i >= total) return;
    const int g = (int)(i % ng);
    const int64_t pix = i / ng;
    const int n = (int)(pix / HW);
    const int64_t hw = pix - (int64_t)n * HW;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        f[e] = (c < C) ? (float)src[((int64_t)n * C + c) * HW + hw] : 0.0f;
    }
    store8<T, DEL>(dst, pix * Cpad + g * 8, 0, f);
}

// NHWC(ld) 16-bit or fp32 -> NCHW. Thread per (n, c, hw) element; hw fastest => coalesced writes.
template <typename DST, typename T>      // T: the source element type (compute type or float)
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, DST* __restrict__ dst, int N, int C, int64_t HW, int ld,
                                    int do_clamp, float lo, float hi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * C * HW;
    if (i >= total) return;
    const int64_t hw = i % HW;
    const int64_t nc = i / HW;
    const int c = (int)(nc % C);
    const int n = (int)(nc / C);
    float v = (float)src[((int64_t)n * HW + hw) * ld + c];
    if (do_clamp) v = fminf(fmaxf(v, lo), hi);
    dst[i] = (DST)v;
}

// all extents in 16-byte chunks (8 x 16-bit or 4 x fp32 channels)
__global__ void copy_channels_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int64_t rows, int ng, int src_ld,
                                     int dst_ld, int dst_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ng) return;
    const int g = (int)(i % ng);
    const int64_t r = i / ng;
    dst[r * dst_ld + dst_off + g] = src[r * src_ld + g];
}

// stream (fp32) -> operand (compute type, plain or two-term split [hi | lo])
template <typename T, int YEL>
__global__ void to_operand_kernel(const float* __restrict__ x, void* __restrict__ y, int64_t rows, int C, unsigned* __restrict__ ovf) {
    const int ng = C >> 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float amax = 0.0f;
    if (i < rows * ng) {
        const int g = (int)(i % ng);
        const int64_t r = i / ng;
        float f[8];
        load8<T, true>(x, r * C + g * 8, f);
        if constexpr (YEL == 3) store8_mx<T>(y, r * 4 * C, C, g * 8, f);
        else if constexpr (YEL == 4) store8_mx6<T>(y, r * 4 * C, C, g * 8, f);        // (lanes 4k .. 4k + 3 = the four octets of one block: ng % 8 == 0)
        else store8<T, YEL>(y, r * (YEL == 2 ? 2 * C : C) + g * 8, C, f);
        if constexpr (std::is_same<T, f16_t>::value) {
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
        }
    }
    if constexpr (std::is_same<T, f16_t>::value) {           // fp16 range guard (see omgsr_igemm_args.overflow_flag)
        if (ovf && __any(amax > 65504.0f) && (threadIdx.x & 63) == 0) atomicOr(ovf, 1u);
        if (YEL == 3 && ovf && __any(amax > 448.0f) && (threadIdx.x & 63) == 0) atomicOr(ovf, 2u);      // MX: correction fields saturated (diagnostic bit)
    }
}

// x f32 [B][L][C] -> y [B][2C][ld]: y[b][c][l] = hi, y[b][C + c][l] = lo of the two-term split, transposed through a 64 x 64 LDS tile
// (reads coalesced along c, 16-byte stores along l). The attention's V^T operand in the range-fallback tier (omgsr_attn_args.vt_lo_off).
template <typename T>
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ x, T* __restrict__ y, int L, int C, int64_t ld) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, l0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const float* xb = x + (int64_t)b * L * C;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int l = i >> 6, c = i & 63;
        tile[l][c] = (l0 + l < L && c0 + c < C) ? xb[(int64_t)(l0 + l) * C + c0 + c] : 0.0f;
    }
    __syncthreads();
    T* yb = y + (int64_t)b * 2 * C * ld;
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {          // 64 channels x 8 groups of 8 consecutive l
        const int c = i >> 3, g = (i & 7) * 8;
        if (c0 + c >= C || l0 + g >= L) continue;
        float hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = tile[g + e][c];
            const T h = (T)v;
            hi[e] = (float)h;
            lo[e] = v - hi[e];
        }
        T* dh = yb + (int64_t)(c0 + c) * ld + l0 + g;
        T* dl = dh + (int64_t)C * ld;
        if (l0 + g + 8 <= L) {
            *reinterpret_cast<u32x4_t*>(dh) = pack8<T>(hi);
            *reinterpret_cast<u32x4_t*>(dl) = pack8<T>(lo);
        } else {
            for (int e = 0; e < 8 && l0 + g + e < L; ++e) { dh[e] = (T)hi[e]; dl[e] = (T)lo[e]; }
        }
    }
}

// z = ((mu + exp(0.5*clamp(logvar,-30,20)) * eps) - shift) * scale, fp32 math, one rounding.
template <typename T>      // T: compute type or float (moments and z share it)
__global__ void vae_sample_kernel(const T* __restrict__ mom, const float* __restrict__ eps, T* __restrict__ z, int64_t rows,
                                  int C, int ld_out, float shift, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ld_out) return;
    const int c = (int)(i % ld_out);
    const int64_t r = i / ld_out;
    float out = 0.0f;
    if (c < C) {
        const float mu = (float)mom[r * 2 * C + c];
        float lv = (float)mom[r * 2 * C + C + c];
        lv = fminf(fmaxf(lv, -30.0f), 20.0f);
        out = ((mu + __expf(0.5f * lv) * eps[r * C + c]) - shift) * scale;
    }
    z[i] = (T)out;
}

template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, int64_t n, float a,
                             float b, float c, float d, int steps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = (float)x[i];
    const float yv = y ? (float)y[i] : 0.0f;
    float v;
    if (steps) {
        const float t0 = (a == 1.0f) ? xv : round_to<T>(xv * a);
        const float t1 = y ? round_to<T>(yv * b) : 0.0f;
        v = y ? round_to<T>(t0 + t1) : t0;
        if (c != 0.0f) v = round_to<T>(v + c);
        if (d != 1.0f) v = round_to<T>(v * d);
    } else {
        v = (xv * a + yv * b + c) * d;
    }
    out[i] = (T)v;
}

// acc[n, y0+y, x0+x, c] += tile[n,y,x,c] * w[y,x]; wsum[n?]: handled by a second call with tile==NULL
template <typename T>
__global__ void tile_accumulate_kernel(const T* __restrict__ tile, const float* __restrict__ w, float* __restrict__ acc, int N,
                                       int C, int th, int tw, int tile_ld, int H, int W, int y0, int x0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * th * tw * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int x = (int)(r % tw); r /= tw;
    const int y = (int)(r % th);
    const int n = (int)(r / th);
    const float wv = w[y * tw + x];
    const float tv = tile ? (float)tile[(((int64_t)n * th + y) * tw + x) * tile_ld + c] : 1.0f;
    acc[(((int64_t)n * H + y0 + y) * W + x0 + x) * C + c] += tv * wv;
}

template <typename T>      // T: compute type or float
__global__ void tile_normalise_kernel(const float* __restrict__ acc, const float* __restrict__ wsum, T* __restrict__ out, int N,
                                      int64_t HW, int C, int ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * HW * ld) return;
    const int c = (int)(i % ld);
    const int64_t pix = i / ld;
    float v = 0.0f;
    if (c < C) v = acc[pix * C + c] / wsum[pix % HW];
    out[i] = (T)v;
}

__global__ void crop_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int N, int H, int W, int ng, int y0, int x0, int th,
                            int tw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * th * tw * ng) return;
    const int g = (int)(i % ng);
    int64_t r = i / ng;
    const int x = (int)(r % tw); r /= tw;
    const int y = (int)(r % th);
    const int n = (int)(r / th);
    dst[i] = src[(((int64_t)n * H + y0 + y) * W + x0 + x) * ng + g];
}

// F.interpolate(mode = "nearest-exact") on an NHWC tensor, 16-byte chunks: src index = min(floorf((dst + 0.5) * scale), in - 1) with
// scale = (float)(1 / scale_factor), the product in double and rounded to float before the floor - ATen's CPU formula (nearest_exact_idx),
// which is what the reference's fast tiled-VAE mode runs (infer/vaehook.py:714-735).
__global__ void resize_nearest_exact_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int N, int H, int W, int ng, int Ho, int Wo,
                                            float sy, float sx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Ho * Wo * ng) return;
    const int g = (int)(i % ng);
    int64_t r = i / ng;
    const int x = (int)(r % Wo); r /= Wo;
    const int y = (int)(r % Ho);
    const int n = (int)(r / Ho);
    int iy = (int)floorf((float)(((double)y + 0.5) * (double)sy)), ix = (int)floorf((float)(((double)x + 0.5) * (double)sx));
    iy = iy < H - 1 ? iy : H - 1;
    ix = ix < W - 1 ? ix : W - 1;
    dst[i] = src[(((int64_t)n * H + iy) * W + ix) * ng + g];
}

__global__ void paste_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int N, int ng, int sH, int sW, int sy0, int sx0,
                             int dH, int dW, int dy0, int dx0, int th, int tw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * th * tw * ng) return;
    const int g = (int)(i % ng);
    int64_t r = i / ng;
    const int x = (int)(r % tw); r /= tw;
    const int y = (int)(r % th);
    const int n = (int)(r / th);
    dst[(((int64_t)n * dH + dy0 + y) * dW + dx0 + x) * ng + g] = src[(((int64_t)n * sH + sy0 + y) * sW + sx0 + x) * ng + g];
}

// Flux 2x2 pack: tokens[n, (y/2)*(W/2) + x/2, c*4 + (y&1)*2 + (x&1)] <-> nhwc[n, y, x, c]
template <typename E>       // E: any 2- or 4-byte element (pure data movement)
__global__ void flux_pack_kernel(const E* __restrict__ src, E* __restrict__ dst, int N, int H, int W, int C, int ld, int dir) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W * C) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    const int64_t nhwc = (((int64_t)n * H + y) * W + x) * ld + c;
    const int64_t tok = (((int64_t)n * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * (4 * C) + c * 4 + (y & 1) * 2 + (x & 1);
    if (dir == 0) dst[tok] = src[nhwc]; else dst[nhwc] = src[tok];
}

inline dim3 grid1d(int64_t total, int block = 256) { return dim3((unsigned)((total + block - 1) / block)); }

}  // namespace

extern "C" int omgsr_abi_version(void) { return 17; }

extern "C" int omgsr_set_compute_dtype(int dtype) {
    if (dtype != OMGSR_DT_BF16 && dtype != OMGSR_DT_F16) return OMGSR_E_BADARG;
    omgsr::g_compute_dtype = dtype;
    return 0;
}
extern "C" int omgsr_get_compute_dtype(void) { return omgsr::g_compute_dtype; }

extern "C" int omgsr_check_device(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return OMGSR_E_ARCH;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return OMGSR_E_ARCH;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 0 : OMGSR_E_ARCH;
}

extern "C" const char* omgsr_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case OMGSR_E_BADARG: return "omgsr: bad argument (null pointer or non-positive dimension)";
        case OMGSR_E_SHAPE: return "omgsr: shape/alignment not supported by the gfx950 kernel";
        case OMGSR_E_ARCH: return "omgsr: current device is not gfx950 (MI355X)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "omgsr: unknown error";
    }
}

extern "C" int omgsr_nchw_to_nhwc(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cpad,
                                  int32_t src_dtype, int32_t dst_el, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0) return OMGSR_E_BADARG;
    if ((Cpad & 7) || Cpad < C) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W, total = (int64_t)N * HW * (Cpad >> 3);
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 2.0 * N * HW * (C + Cpad), st);
    if (dst_el != OMGSR_EL_16 && dst_el != OMGSR_EL_F32) return OMGSR_E_BADARG;
    if (src_dtype == 1 && dst_el == OMGSR_EL_F32) OMGSR_DISPATCH_T(hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, T, 1>), grid1d(total), dim3(256), 0, st, (const float*)src, dst, N, C, HW, Cpad));
    else if (src_dtype == 1) OMGSR_DISPATCH_T(hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, T, 0>), grid1d(total), dim3(256), 0, st, (const float*)src, dst, N, C, HW, Cpad));
    else if (dst_el == OMGSR_EL_F32) OMGSR_DISPATCH_T(hipLaunchKernelGGL((nchw_to_nhwc_kernel<T, T, 1>), grid1d(total), dim3(256), 0, st, (const T*)src, dst, N, C, HW, Cpad));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((nchw_to_nhwc_kernel<T, T, 0>), grid1d(total), dim3(256), 0, st, (const T*)src, dst, N, C, HW, Cpad));
    return (int)hipGetLastError();
}

extern "C" int omgsr_nhwc_to_nchw(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ld,
                                  int32_t dst_dtype, int32_t do_clamp, float lo, float hi, int32_t src_el, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ld < C) return OMGSR_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W, total = (int64_t)N * C * HW;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 4.0 * total, st);
    if (src_el == OMGSR_EL_F32) {
        if (dst_dtype == 1) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, float>), grid1d(total), dim3(256), 0, st, (const float*)src, (float*)dst, N, C, HW, ld, do_clamp, lo, hi);
        else OMGSR_DISPATCH_T(hipLaunchKernelGGL((nhwc_to_nchw_kernel<T, float>), grid1d(total), dim3(256), 0, st, (const float*)src, (T*)dst, N, C, HW, ld, do_clamp, lo, hi));
    } else if (dst_dtype == 1) OMGSR_DISPATCH_T(hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, T>), grid1d(total), dim3(256), 0, st, (const T*)src, (float*)dst, N, C, HW, ld, do_clamp, lo, hi));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((nhwc_to_nchw_kernel<T, T>), grid1d(total), dim3(256), 0, st, (const T*)src, (T*)dst, N, C, HW, ld, do_clamp, lo, hi));
    return (int)hipGetLastError();
}

extern "C" int omgsr_copy_channels(const void* src, void* dst, int64_t rows, int32_t C, int32_t src_ld, int32_t dst_ld,
                                   int32_t dst_off, int32_t el, void* stream) {
    if (!src || !dst || rows <= 0 || C <= 0 || (el != OMGSR_EL_16 && el != OMGSR_EL_F32)) return OMGSR_E_BADARG;
    if ((C & 7) || (src_ld & 7) || (dst_ld & 7) || (dst_off & 7) || dst_off + C > dst_ld || src_ld < C) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int sh = el == OMGSR_EL_F32 ? 2 : 3;          // channels per 16-byte chunk: 4 (fp32) or 8 (16-bit)
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, (el == OMGSR_EL_F32 ? 8.0 : 4.0) * rows * C, st);
    hipLaunchKernelGGL(copy_channels_kernel, grid1d(rows * (C >> sh)), dim3(256), 0, st, (const u32x4_t*)src, (u32x4_t*)dst, rows, C >> sh,
                       src_ld >> sh, dst_ld >> sh, dst_off >> sh);
    return (int)hipGetLastError();
}

extern "C" int omgsr_to_operand(const float* x, void* y, int64_t rows, int32_t C, int32_t y_el, uint32_t* overflow_flag, void* stream) {
    if (!x || !y || rows <= 0 || C <= 0 || (y_el != OMGSR_EL_16 && y_el != OMGSR_EL_SPLIT && y_el != OMGSR_EL_MX && y_el != OMGSR_EL_MX6)) return OMGSR_E_BADARG;
    if (C & 7) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, (y_el == OMGSR_EL_16 ? 6.0 : 8.0) * rows * C, st);
    if (y_el == OMGSR_EL_MX || y_el == OMGSR_EL_MX6) {        // fp16 compute type, whole 64-channel correction chunks
        if ((C & 63) || omgsr::compute_dtype() != 1) return OMGSR_E_SHAPE;
        if (y_el == OMGSR_EL_MX6) hipLaunchKernelGGL((to_operand_kernel<f16_t, 4>), grid1d(rows * (C >> 3)), dim3(256), 0, st, x, y, rows, C, overflow_flag);
        else hipLaunchKernelGGL((to_operand_kernel<f16_t, 3>), grid1d(rows * (C >> 3)), dim3(256), 0, st, x, y, rows, C, overflow_flag);
    } else if (y_el == OMGSR_EL_SPLIT) OMGSR_DISPATCH_T(hipLaunchKernelGGL((to_operand_kernel<T, 2>), grid1d(rows * (C >> 3)), dim3(256), 0, st, x, y, rows, C, overflow_flag));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((to_operand_kernel<T, 0>), grid1d(rows * (C >> 3)), dim3(256), 0, st, x, y, rows, C, overflow_flag));
    return (int)hipGetLastError();
}

extern "C" int omgsr_transpose_split(const float* x, void* y, int32_t B, int32_t L, int32_t C, int64_t ld, void* stream) {
    if (!x || !y || B <= 0 || L <= 0 || C <= 0) return OMGSR_E_BADARG;
    if (ld < L || (ld & 7)) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 8.0 * (double)B * L * C, st);
    const dim3 grid((unsigned)((L + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B);
    OMGSR_DISPATCH_T(hipLaunchKernelGGL(transpose_split_kernel<T>, grid, dim3(256), 0, st, x, (T*)y, L, C, ld));
    return (int)hipGetLastError();
}

extern "C" int omgsr_vae_sample(const void* moments, const float* eps, void* z, int64_t rows, int32_t C, int32_t ld_out,
                                float shift, float scale, int32_t el, void* stream) {
    if (!moments || !eps || !z || rows <= 0 || C <= 0 || ld_out < C) return OMGSR_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 10.0 * rows * C, st);
    if (el == OMGSR_EL_F32) hipLaunchKernelGGL(vae_sample_kernel<float>, grid1d(rows * ld_out), dim3(256), 0, st, (const float*)moments, eps, (float*)z, rows, C, ld_out, shift, scale);
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL(vae_sample_kernel<T>, grid1d(rows * ld_out), dim3(256), 0, st, (const T*)moments, eps, (T*)z, rows, C, ld_out, shift, scale));
    return (int)hipGetLastError();
}

extern "C" int omgsr_axpby(const void* x, const void* y, void* out, int64_t n, float a, float b, float c, float d,
                           int32_t bf16_steps, int32_t el, void* stream) {
    if (!x || !out || n <= 0) return OMGSR_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 6.0 * n, st);
    if (el == OMGSR_EL_F32) hipLaunchKernelGGL(axpby_kernel<float>, grid1d(n), dim3(256), 0, st, (const float*)x, (const float*)y, (float*)out, n, a, b, c, d, 0);
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL(axpby_kernel<T>, grid1d(n), dim3(256), 0, st, (const T*)x, (const T*)y, (T*)out, n, a, b, c, d, bf16_steps));
    return (int)hipGetLastError();
}

namespace {
// y[r][n] = sum_k act(x[r][k]) * w[n][k] + b[n] in fp32, one wave per output element, FIXED summation order (lane-strided partials, then the
// xor butterfly): the constant folds a model runs once per (timestep, prompt) - time embeddings, adaLN modulation vectors, time_emb_proj -
// so that no vendor BLAS kernel runs in the product process. HBM-bound on w (read once, 256-byte coalesced rows).
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                         float* __restrict__ y, int rows, int K, int N, int silu_in) {
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (int64_t)rows * N) return;
    const int r = (int)(o / N), n = (int)(o % N), lane = threadIdx.x & 63;
    const float* xr = x + (int64_t)r * K;
    const float* wn = w + (int64_t)n * K;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv = xr[k];
        if (silu_in) xv = xv / (1.0f + expf(-xv));
        s = fmaf(xv, wn[k], s);
    }
    s = wave_sum(s);
    if (lane == 0) y[o] = s + (b ? b[n] : 0.f);
}
}  // namespace

extern "C" int omgsr_linear_f32(const float* x, const float* w, const float* b, float* y, int32_t rows, int32_t K, int32_t N,
                                int32_t silu_in, void* stream) {
    if (!x || !w || !y || rows <= 0 || K <= 0 || N <= 0) return OMGSR_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 2.0 * rows * K * N, 4.0 * ((double)N * K + (double)rows * (K + N)), st);
    const int64_t outs = (int64_t)rows * N;
    hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, st, x, w, b, y, rows, K, N, silu_in);
    return (int)hipGetLastError();
}

extern "C" int omgsr_tile_accumulate(const void* tile, const float* w, float* acc, int32_t N, int32_t C, int32_t th, int32_t tw,
                                     int32_t tile_ld, int32_t H, int32_t W, int32_t y0, int32_t x0, int32_t tile_el, void* stream) {
    if (!w || !acc || N <= 0 || C <= 0 || th <= 0 || tw <= 0) return OMGSR_E_BADARG;
    if (y0 < 0 || x0 < 0 || y0 + th > H || x0 + tw > W || (tile && tile_ld < C)) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)N * th * tw * C;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 10.0 * total, st);
    if (tile_el == OMGSR_EL_F32) hipLaunchKernelGGL(tile_accumulate_kernel<float>, grid1d(total), dim3(256), 0, st, (const float*)tile, w, acc, N, C, th, tw, tile_ld, H, W, y0, x0);
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL(tile_accumulate_kernel<T>, grid1d(total), dim3(256), 0, st, (const T*)tile, w, acc, N, C, th, tw, tile_ld, H, W, y0, x0));
    return (int)hipGetLastError();
}

extern "C" int omgsr_tile_normalise(const float* acc, const float* wsum, void* out, int32_t N, int64_t HW, int32_t C, int32_t ld,
                                    int32_t out_el, void* stream) {
    if (!acc || !wsum || !out || N <= 0 || HW <= 0 || C <= 0 || ld < C) return OMGSR_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)N * HW * ld;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 6.0 * total, st);
    if (out_el == OMGSR_EL_F32) hipLaunchKernelGGL(tile_normalise_kernel<float>, grid1d(total), dim3(256), 0, st, acc, wsum, (float*)out, N, HW, C, ld);
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL(tile_normalise_kernel<T>, grid1d(total), dim3(256), 0, st, acc, wsum, (T*)out, N, HW, C, ld));
    return (int)hipGetLastError();
}

extern "C" int omgsr_crop_nhwc(const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t y0, int32_t x0,
                               int32_t th, int32_t tw, int32_t el, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || th <= 0 || tw <= 0) return OMGSR_E_BADARG;
    if ((C & 7) || y0 < 0 || x0 < 0 || y0 + th > H || x0 + tw > W) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int ng = el == OMGSR_EL_F32 ? C >> 2 : C >> 3;          // 16-byte chunks per pixel
    const int64_t total = (int64_t)N * th * tw * ng;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 32.0 * total, st);
    hipLaunchKernelGGL(crop_kernel, grid1d(total), dim3(256), 0, st, (const u32x4_t*)src, (u32x4_t*)dst, N, H, W, ng, y0, x0, th, tw);
    return (int)hipGetLastError();
}

extern "C" int omgsr_resize_nearest_exact_nhwc(const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                                               float scale_y, float scale_x, int32_t el, void* stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return OMGSR_E_BADARG;
    if ((el == OMGSR_EL_F32 ? (C & 3) : (C & 7)) || !(scale_y > 0.0f) || !(scale_x > 0.0f)) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int ng = el == OMGSR_EL_F32 ? C >> 2 : C >> 3;
    const int64_t total = (int64_t)N * Ho * Wo * ng;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 32.0 * total, st);
    hipLaunchKernelGGL(resize_nearest_exact_kernel, grid1d(total), dim3(256), 0, st, (const u32x4_t*)src, (u32x4_t*)dst, N, H, W, ng, Ho, Wo, scale_y, scale_x);
    return (int)hipGetLastError();
}

extern "C" int omgsr_paste_nhwc(const void* src, void* dst, int32_t N, int32_t C, int32_t sH, int32_t sW, int32_t sy0, int32_t sx0,
                                int32_t dH, int32_t dW, int32_t dy0, int32_t dx0, int32_t th, int32_t tw, int32_t el, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || th <= 0 || tw <= 0) return OMGSR_E_BADARG;
    if ((C & 7) || sy0 < 0 || sx0 < 0 || dy0 < 0 || dx0 < 0 || sy0 + th > sH || sx0 + tw > sW || dy0 + th > dH || dx0 + tw > dW)
        return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int ng = el == OMGSR_EL_F32 ? C >> 2 : C >> 3;
    const int64_t total = (int64_t)N * th * tw * ng;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 32.0 * total, st);
    hipLaunchKernelGGL(paste_kernel, grid1d(total), dim3(256), 0, st, (const u32x4_t*)src, (u32x4_t*)dst, N, ng, sH, sW, sy0, sx0, dH, dW, dy0, dx0, th, tw);
    return (int)hipGetLastError();
}

extern "C" int omgsr_flux_pack(const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ld, int32_t dir,
                               int32_t el, void* stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || C <= 0 || ld < C) return OMGSR_E_BADARG;
    if ((H & 1) || (W & 1)) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)N * H * W * C;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 4.0 * total, st);
    if (el == OMGSR_EL_F32) hipLaunchKernelGGL(flux_pack_kernel<float>, grid1d(total), dim3(256), 0, st, (const float*)src, (float*)dst, N, H, W, C, ld, dir);
    else hipLaunchKernelGGL(flux_pack_kernel<unsigned short>, grid1d(total), dim3(256), 0, st, (const unsigned short*)src, (unsigned short*)dst, N, H, W, C, ld, dir);
    return (int)hipGetLastError();
}

extern "C" int omgsr_timing_enable(int on) { omgsr::timing_state().on = (on != 0); return 0; }
extern "C" int omgsr_timing_stage(int stage) { omgsr::timing_state().stage = stage; return 0; }

extern "C" int omgsr_timing_reset(void) {
    auto& s = omgsr::timing_state();
    for (auto& r : s.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    s.recs.clear();
    return 0;
}

extern "C" int omgsr_timing_collect(omgsr_timing_entry* out, int cap) {
    auto& s = omgsr::timing_state();
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    int n = 0;
    for (auto& r : s.recs) {
        if (out && n < cap) {
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, r.e0, r.e1);
            out[n].kind = r.kind; out[n].ms = ms; out[n].flops = r.flops; out[n].bytes = r.bytes;
            out[n].m = r.m; out[n].n = r.n; out[n].k = r.k; out[n].variant = r.variant; out[n].stage = r.stage;
        }
        ++n;
    }
    return n;
}
