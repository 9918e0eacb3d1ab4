// GroupNorm / LayerNorm / row-softmax / RMSNorm+RoPE for gfx950. All HBM-bound: 16-byte loads,
// fp32 statistics, wave64 shuffles + LDS for reductions.
#include "common.hip.h"
#include <type_traits>
#include "../../include/omgsr_hip.h"
#include "timing.hip.h"
#include <stdlib.h>

namespace {

constexpr int GN_PPC = 256;   // pixels per statistics chunk (1024 measured 30 % slower: too few blocks in flight)
// ... on big maps. One image per call (round 5) makes the maps 256 ... 4096 pixels: 1 ... 16 chunks of 256 are 1 ... 16 workgroups walking their
// pixels in ~11 dependent load batches (20 us per launch, 76 launches per 128 -> 512 image). Smaller chunks there: a function of HW alone, so the
// partial order - and with it batch invariance - is unchanged for a given image size.
static inline int gn_ppc(const int64_t HW) { return HW >= 32768 ? GN_PPC : (HW >= 8192 ? 128 : (HW >= 2048 ? 32 : 16)); }

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics, pass 1: per (image, pixel-chunk) partial (sum, sumsq) per group.
// x [N][HW][C] bf16.  grid = (nchunk, N), 256 threads.  LDS: 2*C floats.
template <typename T, bool XF32>
__global__ __launch_bounds__(256) void gn_partial_kernel(const void* __restrict__ x, float* __restrict__ partial,
                                                          int64_t HW, int C, int G, int nchunk, int ppc) {
    // LDS: csum[P][C], csq[P][C] — one slot per (pixel lane, channel), reduced in a FIXED order
    // afterwards (no atomics: results are bitwise reproducible and independent of the batch size)
    extern __shared__ __attribute__((aligned(16))) float gn_lds[];
    const int t = threadIdx.x;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int nch8 = C >> 3;
    const int TP = nch8 < 256 ? nch8 : 256;      // threads per pixel
    const int P = 256 / TP;                      // pixels in flight
    float* csum = gn_lds;
    float* csq = gn_lds + P * C;
    const int64_t p0 = (int64_t)chunk * ppc;
    const int npx = (int)((HW - p0) < ppc ? (HW - p0) : ppc);
    if (t < TP * P) {
        const int pl = t / TP;
        const int64_t base = ((int64_t)n * HW + p0) * C;
        for (int c8 = t % TP; c8 < nch8; c8 += TP) {
            float s[8], q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] = 0.0f; q[e] = 0.0f; }
            // 4 independent 16-byte loads in flight per thread (the 1-deep loop was latency-bound)
            int px = pl;
            for (; px + 3 * P < npx; px += 4 * P) {
                float r[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) load8<T, XF32>(x, base + (int64_t)(px + u * P) * C + c8 * 8, r[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s[e] += r[u][e]; q[e] += r[u][e] * r[u][e]; }
                }
            }
            for (; px < npx; px += P) {
                float f[8];
                load8<T, XF32>(x, base + (int64_t)px * C + c8 * 8, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                csum[pl * C + c8 * 8 + e] = s[e];
                csq[pl * C + c8 * 8 + e] = q[e];
            }
        }
    }
    __syncthreads();
    const int cpg = C / G;
    if (t < G) {
        float s = 0.0f, q = 0.0f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c)
            for (int pl = 0; pl < P; ++pl) { s += csum[pl * C + c]; q += csq[pl * C + c]; }
        float* o = partial + (((int64_t)n * nchunk + chunk) * G + t) * 2;
        o[0] = s; o[1] = q;
    }
}

// pass 2: one thread per (n, g): fold the chunk partials in double.
__global__ void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ var_out, int N, int G, int nchunk, double count, float eps, int entries) {
    // one wave per (n, g): lanes stride the chunk list, fixed-order butterfly in double
    const int i = blockIdx.x;
    const int lane = threadIdx.x;
    const int n = i / G, g = i - n * G;
    double s = 0.0, q = 0.0;
    const int epg = entries / G;           // entries per group: 1, or the group's channels when the producer wrote per channel
    for (int c = lane; c < nchunk; c += 64) {
        const float* p = partial + (((int64_t)n * nchunk + c) * entries + g * epg) * 2;
        for (int j = 0; j < epg; ++j) { s += (double)p[2 * j]; q += (double)p[2 * j + 1]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (lane != 0) return;
    const double m = s / count;
    double v = q / count - m * m;
    if (v < 0.0) v = 0.0;
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(v + (double)eps));
    if (var_out) var_out[i] = (float)v;
}

// GroupNorm apply (+ optional SiLU). grid = (nblk, N); each block builds the per-channel
// (scale, shift) table of its image in LDS, then streams its pixel range.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int64_t HW, int C, int G, int act, int64_t px_per_block, int stat_rows) {
    extern __shared__ __attribute__((aligned(16))) float gn_lds[];
    float* sc = gn_lds;
    float* sh = gn_lds + C;
    const int t = threadIdx.x, n = blockIdx.y;
    const int ns = n % stat_rows;          // tiled VAE: rows are (tile, image) tile-major and share the image's statistics
    const int cpg = C / G;
    for (int c = t; c < C; c += 256) {
        const int g = c / cpg;
        const float r = rstd[ns * G + g], m = mean[ns * G + g];
        const float a = r * (gamma ? gamma[c] : 1.0f);
        sc[c] = a;
        sh[c] = (beta ? beta[c] : 0.0f) - m * a;
    }
    __syncthreads();
    const int nch8 = C >> 3;
    const int64_t p0 = (int64_t)blockIdx.x * px_per_block;
    int64_t p1 = p0 + px_per_block; if (p1 > HW) p1 = HW;
    const int total = (int)((p1 - p0) * nch8);   // <= ~4k chunks per block by construction
    const T* xb = x + ((int64_t)n * HW + p0) * C;
    T* yb = y + ((int64_t)n * HW + p0) * C;
    // two 16-byte chunks in flight per thread; the channel-octet index advances by (256 mod nch8) per step
    const int step8 = 256 % nch8;
    int c8 = t % nch8;
    int i = t;
    for (; i + 256 < total; i += 512) {
        const u32x4_t r0 = *reinterpret_cast<const u32x4_t*>(xb + (int64_t)i * 8);
        const u32x4_t r1 = *reinterpret_cast<const u32x4_t*>(xb + (int64_t)(i + 256) * 8);
        int c8b = c8 + step8; if (c8b >= nch8) c8b -= nch8;
        float f[8], h[8];
        unpack8<T>(r0, f); unpack8<T>(r1, h);
        // the (scale, shift) octets as four ds_read_b128 each (the compiler leaves them as 32 scalar LDS reads)
        const f32x4_t* sa = reinterpret_cast<const f32x4_t*>(sc + c8 * 8);
        const f32x4_t* ha = reinterpret_cast<const f32x4_t*>(sh + c8 * 8);
        const f32x4_t* sb = reinterpret_cast<const f32x4_t*>(sc + c8b * 8);
        const f32x4_t* hb = reinterpret_cast<const f32x4_t*>(sh + c8b * 8);
        const f32x4_t sa0 = sa[0], sa1 = sa[1], ha0 = ha[0], ha1 = ha[1], sb0 = sb[0], sb1 = sb[1], hb0 = hb[0], hb1 = hb[1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = f[e] * (e < 4 ? sa0[e & 3] : sa1[e & 3]) + (e < 4 ? ha0[e & 3] : ha1[e & 3]);
            const float w = h[e] * (e < 4 ? sb0[e & 3] : sb1[e & 3]) + (e < 4 ? hb0[e & 3] : hb1[e & 3]);
            f[e] = (act == OMGSR_ACT_SILU) ? silu_f(v) : v;
            h[e] = (act == OMGSR_ACT_SILU) ? silu_f(w) : w;
        }
        *reinterpret_cast<u32x4_t*>(yb + (int64_t)i * 8) = pack8<T>(f);
        *reinterpret_cast<u32x4_t*>(yb + (int64_t)(i + 256) * 8) = pack8<T>(h);
        c8 = c8b + step8; if (c8 >= nch8) c8 -= nch8;
    }
    for (; i < total; i += 256) {
        float f[8];
        unpack8<T>(*reinterpret_cast<const u32x4_t*>(xb + (int64_t)i * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = f[e] * sc[c8 * 8 + e] + sh[c8 * 8 + e];
            f[e] = (act == OMGSR_ACT_SILU) ? silu_f(v) : v;
        }
        *reinterpret_cast<u32x4_t*>(yb + (int64_t)i * 8) = pack8<T>(f);
        c8 += step8; if (c8 >= nch8) c8 -= nch8;
    }
}

// The same for the accurate tier's element kinds: fp32 stream input (XF32) and / or a two-term split operand output
// (YEL = OMGSR_EL_SPLIT: hi at channel c, lo at channel C + c of a 2C-wide row). Chunk i of the block is (pixel, octet) =
// (i / nch8, i % nch8), tracked incrementally.
// Y2EL >= 0: a second output y2 = x itself as an MFMA operand (plain or two-term split), no affine / activation: the block's
// 1x1 shortcut conv reads the same tensor this kernel already streams, so its cast costs a write and no extra read.
template <typename T, bool XF32, int YEL, int Y2EL = -1>
__global__ __launch_bounds__(256) void gn_apply_any_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int64_t HW, int C, int G, int act, int64_t px_per_block, int stat_rows,
                                                            void* __restrict__ y2 = nullptr, unsigned* __restrict__ ovf = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float gn_lds[];
    float amax = 0.0f;          // fp16 range guard of the raw cast y2 (see omgsr_igemm_args.overflow_flag)
    auto note8 = [&](const float (&v)[8]) {
        if constexpr (Y2EL >= 0 && std::is_same<T, f16_t>::value) {
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
        }
    };
    float* sc = gn_lds;
    float* sh = gn_lds + C;
    const int t = threadIdx.x, n = blockIdx.y;
    const int ns = n % stat_rows;
    const int cpg = C / G;
    for (int c = t; c < C; c += 256) {
        const int g = c / cpg;
        const float r = rstd[ns * G + g], m = mean[ns * G + g];
        const float a = r * (gamma ? gamma[c] : 1.0f);
        sc[c] = a;
        sh[c] = (beta ? beta[c] : 0.0f) - m * a;
    }
    __syncthreads();
    const int nch8 = C >> 3;
    const int64_t p0 = (int64_t)blockIdx.x * px_per_block;
    int64_t p1 = p0 + px_per_block; if (p1 > HW) p1 = HW;
    const int total = (int)((p1 - p0) * nch8);
    const int64_t pix0 = (int64_t)n * HW + p0;
    const int step8 = 256 % nch8, stepp = 256 / nch8;
    const int ldy = YEL == 2 ? 2 * C : C;
    int c8 = t % nch8, px = t / nch8;
    auto advance = [&](int& cc, int& pp) {
        cc += step8; pp += stepp;
        if (cc >= nch8) { cc -= nch8; ++pp; }
    };
    auto transform = [&](float (&f)[8], const int cc) {
        const f32x4_t* sa = reinterpret_cast<const f32x4_t*>(sc + cc * 8);
        const f32x4_t* ha = reinterpret_cast<const f32x4_t*>(sh + cc * 8);
        const f32x4_t sa0 = sa[0], sa1 = sa[1], ha0 = ha[0], ha1 = ha[1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = f[e] * (e < 4 ? sa0[e & 3] : sa1[e & 3]) + (e < 4 ? ha0[e & 3] : ha1[e & 3]);
            f[e] = (act == OMGSR_ACT_SILU) ? silu_f(v) : v;
        }
    };
    // YEL 3 = OMGSR_EL_MX: [hi fp16 | lo' fp8 | hi' fp8], 4C bytes per pixel (common.hip.h store8_mx)
    auto put = [&](const int64_t pix, const int cc8, const float (&v)[8]) {
        if constexpr (YEL == 3) store8_mx<T>(y, pix * 4 * C, C, cc8 * 8, v);
        else if constexpr (YEL == 4) store8_mx6<T>(y, pix * 4 * C, C, cc8 * 8, v);     // OMGSR_EL_MX6; cooperative over lane quads: consecutive lanes hold
                                                                                       // consecutive octets of one pixel and C % 64 == 0 (every loop below keeps quads together)
        else store8<T, YEL>(y, pix * ldy + cc8 * 8, C, v);
    };
    // second output: x itself as an operand - plain, two-term split, or (Y2EL 3, round 4) the mixed-precision form of a 1x1 shortcut
    // conv that runs as an MX GEMM
    auto put2 = [&](const int64_t pix, const int cc8, const float (&v)[8]) {
        if constexpr (Y2EL == 3) store8_mx<T>(y2, pix * 4 * C, C, cc8 * 8, v);
        else if constexpr (Y2EL >= 0) store8<T, Y2EL>(y2, pix * (Y2EL == 2 ? 2 * C : C) + cc8 * 8, C, v);
    };
    // two chunks (4 x 16-byte loads of an fp32 row) in flight per thread
    int i = t;
    for (; i + 256 < total; i += 512) {
        int c8b = c8, pxb = px;
        advance(c8b, pxb);
        float f[8], h[8];
        load8<T, XF32>(x, (pix0 + px) * C + c8 * 8, f);
        load8<T, XF32>(x, (pix0 + pxb) * C + c8b * 8, h);
        if constexpr (Y2EL >= 0) {
            note8(f); note8(h);
            put2(pix0 + px, c8, f);
            put2(pix0 + pxb, c8b, h);
        }
        transform(f, c8);
        transform(h, c8b);
        put(pix0 + px, c8, f);
        put(pix0 + pxb, c8b, h);
        c8 = c8b; px = pxb;
        advance(c8, px);
    }
    for (; i < total; i += 256) {
        float f[8];
        load8<T, XF32>(x, (pix0 + px) * C + c8 * 8, f);
        if constexpr (Y2EL >= 0) { note8(f); put2(pix0 + px, c8, f); }
        transform(f, c8);
        put(pix0 + px, c8, f);
        advance(c8, px);
    }
    if constexpr (Y2EL >= 0 && std::is_same<T, f16_t>::value) {
        if (ovf && __any(amax > 65504.0f) && (t & 63) == 0) atomicOr(ovf, 1u);
        if (Y2EL == 3 && ovf && __any(amax > 448.0f) && (t & 63) == 0) atomicOr(ovf, 2u);     // MX twin: correction fields saturated (diagnostic bit)
    }
}

// (scale, shift) table of a GroupNorm for the conv kernels that normalise while they build their patch (omgsr_igemm_args.gn_scale_shift):
// the same two expressions the apply kernels above evaluate per block
__global__ void gn_scale_shift_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float* __restrict__ out, int total, int C, int G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = i / C, c = i - n * C, g = c / (C / G);
    const float r = rstd[n * G + g], m = mean[n * G + g];
    const float a = r * (gamma ? gamma[c] : 1.0f);
    *reinterpret_cast<f32x2_t*>(out + 2 * (int64_t)i) = (f32x2_t){a, (beta ? beta[c] : 0.0f) - m * a};
}
// ---------------------------------------------------------------------------------------------
OMGSR_DEVINL void load_affine8(const float* __restrict__ v, const int c, const float dflt, float (&o)[8]) {
    if (v) {
        const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(v + c), hi = *reinterpret_cast<const f32x4_t*>(v + c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = dflt;
    }
}

// LayerNorm: a wave owns RPW consecutive rows, C <= 64*8*MAXCH. y = (x-mu)*rstd*a[c] + b[c].
// All RPW*MAXCH 16-byte loads of a wave are issued before the first reduction (one row per wave kept a single load
// in flight: 1.05 TB/s on the UNet's [147456, 320] token matrices), the RPW butterfly chains interleave, and the
// affine vectors are fetched once per wave.
template <typename T, int MAXCH, int RPW, bool XF32 = false, int YEL = 0>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                         const float* __restrict__ a, const float* __restrict__ b,
                                                         int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int nch8 = C >> 3;
    float f[RPW][MAXCH][8];
    float s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        s[r] = 0.0f;
        const bool live = row0 + r < rows;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c8 = lane + 64 * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[r][i][e] = 0.0f;
            if (live && c8 < nch8) load8<T, XF32>(x, (row0 + r) * C + c8 * 8, f[r][i]);
        }
    }
    constexpr bool PRE = MAXCH <= 3;          // wide rows: the affine vectors would cost 16*MAXCH registers, fetch them at use
    float av[PRE ? MAXCH : 1][8], bv[PRE ? MAXCH : 1][8];
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c8 = lane + 64 * i;
            load_affine8(c8 < nch8 ? a : nullptr, c8 * 8, 1.0f, av[i]);
            load_affine8(c8 < nch8 ? b : nullptr, c8 * 8, 0.0f, bv[i]);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[r] += f[r][i][e];         // lanes past the row hold zeros
    float mu[RPW], q[RPW];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < RPW; ++r) s[r] += __shfl_xor(s[r], o);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        mu[r] = s[r] / (float)C;
        q[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (lane + 64 * i < nch8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[r][i][e] - mu[r]; q[r] += d * d; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < RPW; ++r) q[r] += __shfl_xor(q[r], o);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        if (row0 + r >= rows) break;
        const float rs = rsqrtf(q[r] / (float)C + eps);
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c8 = lane + 64 * i;
            if (c8 < nch8) {
                float o8[8], ga[8], be[8];
                if constexpr (!PRE) { load_affine8(a, c8 * 8, 1.0f, ga); load_affine8(b, c8 * 8, 0.0f, be); }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o8[e] = (f[r][i][e] - mu[r]) * rs * (PRE ? av[PRE ? i : 0][e] : ga[e]) + (PRE ? bv[PRE ? i : 0][e] : be[e]);
                if constexpr (YEL == 3) {
                    if constexpr (std::is_same<T, f16_t>::value) store8_mx<T>(y, (row0 + r) * 4 * (int64_t)C, C, c8 * 8, o8);   // OMGSR_EL_MX row: 4C bytes
                } else store8<T, YEL>(y, (row0 + r) * (YEL == 2 ? 2 * C : C) + c8 * 8, C, o8);
            }
        }
    }
}

// Wide rows (C >= 1024: UNet level 2, every Flux LayerNorm): one 256-thread block per row, <= MAXV 16-byte chunks per
// thread all in flight at once, block reductions through LDS.
template <typename T, int MAXV, bool XF32 = false, int YEL = 0>
__global__ __launch_bounds__(256) void layernorm_block_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                               const float* __restrict__ a, const float* __restrict__ b,
                                                               int C, float eps) {
    __shared__ float red[2][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t row = blockIdx.x;
    const int nch8 = C >> 3;
    float f[MAXV][8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c8 = t + 256 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[i][e] = 0.0f;
        if (c8 < nch8) load8<T, XF32>(x, row * C + c8 * 8, f[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[i][e];
    }
    s = wave_sum(s);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    const float mu = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (t + 256 * i < nch8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mu; q += d * d; }
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[1][wave] = q;
    __syncthreads();
    const float rs = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c8 = t + 256 * i;
        if (c8 < nch8) {
            float o8[8], ga[8], be[8];
            load_affine8(a, c8 * 8, 1.0f, ga); load_affine8(b, c8 * 8, 0.0f, be);
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = (f[i][e] - mu) * rs * ga[e] + be[e];
            if constexpr (YEL == 3) {
                if constexpr (std::is_same<T, f16_t>::value) store8_mx<T>(y, row * 4 * (int64_t)C, C, c8 * 8, o8);
            } else store8<T, YEL>(y, row * (YEL == 2 ? 2 * C : C) + c8 * 8, C, o8);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row softmax fp32 -> bf16, one 256-thread block per row, L <= 256*4*MAXV.
// SPLIT (round 6): the probabilities as the two-term split [p_hi (L) | p_lo (L)] per row (range-fallback tier: the PV product of the VAE's
// one-head attention runs p_hi v_hi + p_lo v_hi + p_hi v_lo, bmm_nt(both_split)).
template <typename T, int MAXV, bool SPLIT = false>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, T* __restrict__ p, int L, int Lvalid) {
    __shared__ float red[8];
    const int t = threadIdx.x;
    const float* sr = s + (int64_t)blockIdx.x * L;
    T* pr = p + (int64_t)blockIdx.x * (SPLIT ? 2 * L : L);
    const int nv = L >> 2;
    f32x4_t v[MAXV];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = t + 256 * i;
        if (j < nv) {
            v[i] = *reinterpret_cast<const f32x4_t*>(sr + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (4 * j + e >= Lvalid) v[i][e] = -INFINITY;   // padded keys
            mx = fmaxf(mx, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
        }
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = t + 256 * i;
        if (j < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] = __expf(v[i][e] - mx); sum += v[i][e]; }
        }
    }
    sum = wave_sum(sum);
    if ((t & 63) == 0) red[4 + (t >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = t + 256 * i;
        if (j < nv) {
            u32x2_t o;
            o[0] = pack2<T>(v[i][0] * inv, v[i][1] * inv);
            o[1] = pack2<T>(v[i][2] * inv, v[i][3] * inv);
            *reinterpret_cast<u32x2_t*>(pr + 4 * j) = o;
            if constexpr (SPLIT) {
                const T* hp = reinterpret_cast<const T*>(&o);
                u32x2_t l;
                l[0] = pack2<T>(v[i][0] * inv - (float)hp[0], v[i][1] * inv - (float)hp[1]);
                l[1] = pack2<T>(v[i][2] * inv - (float)hp[2], v[i][3] * inv - (float)hp[3]);
                *reinterpret_cast<u32x2_t*>(pr + L + 4 * j) = l;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// RMSNorm over head_dim (D = 128) * w, then interleaved-pair RoPE, in place.  16 lanes per head
// (8 elements each), 4 heads per wave.
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ w2, int Lsplit,
                                                            const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                            int64_t rows, int L, int H, int D, int64_t ld, int col0,
                                                            int pos0, float eps) {
    const int64_t gid = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);   // (row, head) index
    if (gid >= rows * H) return;
    const int sub = threadIdx.x & 15;
    const int64_t row = gid / H;
    const int h = (int)(gid - row * H);
    T* px = x + row * ld + col0 + h * D + sub * 8;
    float f[8];
    unpack8<T>(*reinterpret_cast<const u32x4_t*>(px), f);
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) q += f[e] * f[e];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float r = rsqrtf(q / (float)D + eps);
    const float* wt = (w2 && (int)(row % L) >= Lsplit) ? w2 : w;       // joint [text ; image] sequence: two weight tables
    // weight / cos / sin rows as 2 x 16-byte loads each (24 scalar loads per thread made this pass run at 4.3 TB/s)
    float wv[8];
    load_affine8(wt, h * D + sub * 8, 1.0f, wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = f[e] * r * wv[e];
    if (cos_t) {
        const int pos = pos0 + (int)(row % L);
        float cp[8], sp[8];
        load_affine8(cos_t, (int)((int64_t)pos * D) + sub * 8, 1.0f, cp);
        load_affine8(sin_t, (int)((int64_t)pos * D) + sub * 8, 0.0f, sp);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float a = f[e], b = f[e + 1];
            f[e] = a * cp[e] - b * sp[e];
            f[e + 1] = b * cp[e + 1] + a * sp[e + 1];
        }
    }
    *reinterpret_cast<u32x4_t*>(px) = pack8<T>(f);
}

}  // namespace

extern "C" int omgsr_groupnorm_scale_shift(const float* mean, const float* rstd, const float* gamma, const float* beta, float* out, int32_t nimg,
                                           int32_t C, int32_t G, void* stream) {
    if (!mean || !rstd || !out || nimg <= 0 || C <= 0 || G <= 0 || (C % G)) return OMGSR_E_BADARG;
    const int total = nimg * C;
    hipLaunchKernelGGL(gn_scale_shift_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, rstd, gamma, beta, out, total, C, G);
    return (int)hipGetLastError();
}

extern "C" int omgsr_groupnorm_nchunk(int64_t HW) { const int ppc = gn_ppc(HW); return (int)((HW + ppc - 1) / ppc); }

extern "C" int omgsr_groupnorm_stats(const void* x, float* partial, float* mean, float* rstd, float* var_out,
                                     int32_t N, int64_t HW, int32_t C, int32_t G, float eps, int32_t x_el, void* stream) {
    if (!x || !partial || !mean || !rstd || N <= 0 || HW <= 0 || C <= 0 || G <= 0) return OMGSR_E_BADARG;
    if ((C & 7) || (C % G) || G > 256 || C > 8192) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = omgsr_groupnorm_nchunk(HW);
    omgsr::TimingScope ts(OMGSR_TK_GN, 0.0, (x_el == OMGSR_EL_F32 ? 4.0 : 2.0) * N * (double)HW * C, st);
    const int tp = (C >> 3) < 256 ? (C >> 3) : 256;
    const size_t lds = 2 * (size_t)(256 / tp) * C * sizeof(float);      // <= 20 KB
    if (x_el == OMGSR_EL_F32) OMGSR_DISPATCH_T(hipLaunchKernelGGL((gn_partial_kernel<T, true>), dim3(nchunk, N), dim3(256), lds, st, x, partial, HW, C, G, nchunk, gn_ppc(HW)));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((gn_partial_kernel<T, false>), dim3(nchunk, N), dim3(256), lds, st, x, partial, HW, C, G, nchunk, gn_ppc(HW)));
    const int tot = N * G;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(tot), dim3(64), 0, st, partial, mean, rstd, var_out,
                       N, G, nchunk, (double)HW * (C / G), eps, G);
    return (int)hipGetLastError();
}

extern "C" int omgsr_groupnorm_finalize(const float* partial, float* mean, float* rstd, float* var_out, int32_t N,
                                        int32_t nslot, int32_t G, int32_t entries, double count, float eps, void* stream) {
    if (!partial || !mean || !rstd || N <= 0 || nslot <= 0 || G <= 0 || count <= 0.0 || entries < G || (entries % G)) return OMGSR_E_BADARG;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * G), dim3(64), 0, (hipStream_t)stream, partial, mean, rstd, var_out,
                       N, G, nslot, count, eps, entries);
    return (int)hipGetLastError();
}

namespace {
// pass 2 for the tiled VAE: per-tile (mean, var) of every shape group, folded per image with pixel-count weights
// (infer/vaehook.py GroupNormParam.summary: mean = sum_t w_t mean_t, var = sum_t w_t var_t). One wave per (n, g).
__global__ __launch_bounds__(256) void gn_finalize_merged_kernel(const omgsr_gn_merge_args a, float* __restrict__ mean, float* __restrict__ rstd,
                                                                  float* __restrict__ var_out, int N, int G, float eps) {
    // one 256-thread block per (n, g); the (group, tile) pairs go round-robin to the four waves, each wave folds its
    // tile's slots with 64 lanes; the four weighted partial results are combined in wave order (deterministic)
    __shared__ double red[4][2];
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = i / G, g = i - n * G;
    double m_acc = 0.0, v_acc = 0.0;
    int pair = 0;
    for (int k = 0; k < a.ngroups; ++k) {
        const float* partial = a.partial[k];
        const int nslot = a.nslot[k];
        for (int t = 0; t < a.tiles[k]; ++t, ++pair) {
            if ((pair & 3) != wave) continue;
            const int row = t * N + n;
            double s = 0.0, q = 0.0;
            const int entries = a.entries[k], epg = entries / G;
            for (int c = lane; c < nslot; c += 64) {
                const float* pp = partial + (((int64_t)row * nslot + c) * entries + g * epg) * 2;
                for (int j = 0; j < epg; ++j) { s += (double)pp[2 * j]; q += (double)pp[2 * j + 1]; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
            const double m = s / a.count[k];
            double v = q / a.count[k] - m * m;
            if (v < 0.0) v = 0.0;
            m_acc += (double)a.weight[k] * m;
            v_acc += (double)a.weight[k] * v;
        }
    }
    if (lane == 0) { red[wave][0] = m_acc; red[wave][1] = v_acc; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double mm = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
    const double vv = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    mean[i] = (float)mm;
    rstd[i] = (float)(1.0 / sqrt(vv + (double)eps));
    if (var_out) var_out[i] = (float)vv;
}

}  // namespace
namespace {
int gn_apply_launch(const void* x, void* y, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t act, int32_t stat_rows, int32_t x_el,
                    int32_t y_el, void* y2, int32_t y2_el, uint32_t* ovf, void* stream);
}

extern "C" int omgsr_groupnorm_partial(const void* x, float* partial, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t x_el, void* stream) {
    if (!x || !partial || N <= 0 || HW <= 0 || C <= 0 || G <= 0) return OMGSR_E_BADARG;
    if ((C & 7) || (C % G) || G > 256 || C > 8192) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = omgsr_groupnorm_nchunk(HW);
    omgsr::TimingScope ts(OMGSR_TK_GN, 0.0, (x_el == OMGSR_EL_F32 ? 4.0 : 2.0) * N * (double)HW * C, st);
    const int tp = (C >> 3) < 256 ? (C >> 3) : 256;
    const size_t lds = 2 * (size_t)(256 / tp) * C * sizeof(float);
    if (x_el == OMGSR_EL_F32) OMGSR_DISPATCH_T(hipLaunchKernelGGL((gn_partial_kernel<T, true>), dim3(nchunk, N), dim3(256), lds, st, x, partial, HW, C, G, nchunk, gn_ppc(HW)));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((gn_partial_kernel<T, false>), dim3(nchunk, N), dim3(256), lds, st, x, partial, HW, C, G, nchunk, gn_ppc(HW)));
    return (int)hipGetLastError();
}

extern "C" int omgsr_groupnorm_finalize_merged(const omgsr_gn_merge_args* a, float* mean, float* rstd, float* var_out,
                                               int32_t N, int32_t G, float eps, void* stream) {
    if (!a || !mean || !rstd || N <= 0 || G <= 0 || a->ngroups <= 0 || a->ngroups > OMGSR_GN_MAX_GROUPS) return OMGSR_E_BADARG;
    for (int k = 0; k < a->ngroups; ++k)
        if (!a->partial[k] || a->tiles[k] <= 0 || a->nslot[k] <= 0 || a->count[k] <= 0.0 || a->entries[k] < G || (a->entries[k] % G)) return OMGSR_E_BADARG;
    hipLaunchKernelGGL(gn_finalize_merged_kernel, dim3(N * G), dim3(256), 0, (hipStream_t)stream, *a, mean, rstd, var_out, N, G, eps);
    return (int)hipGetLastError();
}

extern "C" int omgsr_groupnorm_apply_shared(const void* x, void* y, const float* mean, const float* rstd, const float* gamma,
                                            const float* beta, int32_t rows, int64_t HW, int32_t C, int32_t G, int32_t act,
                                            int32_t stat_rows, int32_t x_el, int32_t y_el, void* y2, int32_t y2_el, uint32_t* overflow_flag,
                                            void* stream) {
    if (stat_rows <= 0 || rows % stat_rows) return OMGSR_E_BADARG;
    return gn_apply_launch(x, y, mean, rstd, gamma, beta, rows, HW, C, G, act, stat_rows, x_el, y_el, y2, y2_el, overflow_flag, stream);
}

extern "C" int omgsr_groupnorm_apply(const void* x, void* y, const float* mean, const float* rstd, const float* gamma,
                                     const float* beta, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t act,
                                     int32_t x_el, int32_t y_el, void* y2, int32_t y2_el, uint32_t* overflow_flag, void* stream) {
    return gn_apply_launch(x, y, mean, rstd, gamma, beta, N, HW, C, G, act, N, x_el, y_el, y2, y2_el, overflow_flag, stream);
}

namespace {
int gn_apply_launch(const void* x, void* y, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t act, int32_t stat_rows, int32_t x_el,
                    int32_t y_el, void* y2, int32_t y2_el, uint32_t* ovf, void* stream) {
    if (!x || !y || !mean || !rstd || N <= 0 || HW <= 0 || C <= 0 || G <= 0) return OMGSR_E_BADARG;
    if ((x_el != OMGSR_EL_16 && x_el != OMGSR_EL_F32) || (y_el != OMGSR_EL_16 && y_el != OMGSR_EL_SPLIT && y_el != OMGSR_EL_MX && y_el != OMGSR_EL_MX6)) return OMGSR_E_BADARG;
    if ((y_el == OMGSR_EL_MX || y_el == OMGSR_EL_MX6) && ((C & 63) || omgsr::compute_dtype() != 1)) return OMGSR_E_SHAPE;        // fp16 compute type, whole 64-channel correction chunks
    if (y2 && (x_el != OMGSR_EL_F32 || (y2_el != OMGSR_EL_16 && y2_el != OMGSR_EL_SPLIT && y2_el != OMGSR_EL_MX))) return OMGSR_E_BADARG;
    if (y2 && y2_el == OMGSR_EL_MX && ((C & 63) || omgsr::compute_dtype() != 1)) return OMGSR_E_SHAPE;
    if ((C & 7) || (C % G) || C > 8192) return OMGSR_E_SHAPE;
    if (act != OMGSR_ACT_NONE && act != OMGSR_ACT_SILU) return OMGSR_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    // ~64 KB of activations per block keeps >= 2k blocks in flight on the big VAE maps
    static const char* ppe = getenv("OMGSR_GN_BLOCK_ELEMS");           // A/B runs
    static const int64_t belems = ppe ? atol(ppe) : 16384;
    int64_t ppb = (belems + C - 1) / C;      // ~32 KB of activations per block (5.6 TB/s; 64 KB 5.4, 128 KB 4.9, 256 KB 30 % slower)
    if (ppb < 1) ppb = 1;
    // small maps (one image per call): at least ~512 blocks, down to 8 pixels each, instead of 10 ... 80 blocks on 256 CUs
    if (((HW + ppb - 1) / ppb) * N < 512) {
        ppb = (HW * N + 511) / 512;
        if (ppb < 8) ppb = 8;
    }
    const int nblk = (int)((HW + ppb - 1) / ppb);
    omgsr::TimingScope ts(OMGSR_TK_GN, 0.0, ((x_el == OMGSR_EL_F32 ? 4.0 : 2.0) + (y_el == OMGSR_EL_16 ? 2.0 : 4.0) +
                                             (y2 ? (y2_el == OMGSR_EL_16 ? 2.0 : 4.0) : 0.0)) * N * (double)HW * C, st);
    const size_t lds = 2 * C * sizeof(float);
    const dim3 grid(nblk, N);
#define OMGSR_GN_ANY2(YE, Y2) OMGSR_DISPATCH_T(hipLaunchKernelGGL((gn_apply_any_kernel<T, true, YE, Y2>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, y2, ovf))
#define OMGSR_GN_ANY(XF, YE) OMGSR_DISPATCH_T(hipLaunchKernelGGL((gn_apply_any_kernel<T, XF, YE>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows))
    if (y_el == OMGSR_EL_MX6) {              // fp6 correction thirds for a 3x3 conv of the halo-tile kernel (round 5); the twin keeps its own form
        using T = f16_t;
#define OMGSR_GN_6(XF, Y2) hipLaunchKernelGGL((gn_apply_any_kernel<T, XF, 4, Y2>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, y2, ovf)
        if (y2 && y2_el == OMGSR_EL_MX) OMGSR_GN_6(true, 3);
        else if (y2 && y2_el == OMGSR_EL_SPLIT) OMGSR_GN_6(true, 2);
        else if (y2) OMGSR_GN_6(true, 0);
        else if (x_el == OMGSR_EL_F32) hipLaunchKernelGGL((gn_apply_any_kernel<T, true, 4>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, nullptr, nullptr);
        else hipLaunchKernelGGL((gn_apply_any_kernel<T, false, 4>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, nullptr, nullptr);
#undef OMGSR_GN_6
    } else if (y2 && y2_el == OMGSR_EL_MX) {        // the shortcut's operand in the mixed-precision form (fp16 compute type)
        using T = f16_t;
        if (y_el == OMGSR_EL_MX) hipLaunchKernelGGL((gn_apply_any_kernel<T, true, 3, 3>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, y2, ovf);
        else if (y_el == OMGSR_EL_SPLIT) hipLaunchKernelGGL((gn_apply_any_kernel<T, true, 2, 3>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, y2, ovf);
        else hipLaunchKernelGGL((gn_apply_any_kernel<T, true, 0, 3>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, y2, ovf);
    } else if (y_el == OMGSR_EL_MX) {
        using T = f16_t;
        if (y2 && y2_el == OMGSR_EL_SPLIT) hipLaunchKernelGGL((gn_apply_any_kernel<T, true, 3, 2>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, y2, ovf);
        else if (y2) hipLaunchKernelGGL((gn_apply_any_kernel<T, true, 3, 0>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, y2, ovf);
        else if (x_el == OMGSR_EL_F32) hipLaunchKernelGGL((gn_apply_any_kernel<T, true, 3>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, nullptr, nullptr);
        else hipLaunchKernelGGL((gn_apply_any_kernel<T, false, 3>), grid, dim3(256), lds, st, x, y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows, nullptr, nullptr);
    } else if (y2) {
        if (y_el == OMGSR_EL_SPLIT && y2_el == OMGSR_EL_SPLIT) OMGSR_GN_ANY2(2, 2);
        else if (y_el == OMGSR_EL_SPLIT) OMGSR_GN_ANY2(2, 0);
        else if (y2_el == OMGSR_EL_SPLIT) OMGSR_GN_ANY2(0, 2);
        else OMGSR_GN_ANY2(0, 0);
    } else if (x_el == OMGSR_EL_F32 && y_el == OMGSR_EL_SPLIT) OMGSR_GN_ANY(true, 2);
    else if (x_el == OMGSR_EL_F32) OMGSR_GN_ANY(true, 0);
    else if (y_el == OMGSR_EL_SPLIT) OMGSR_GN_ANY(false, 2);
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL(gn_apply_kernel<T>, grid, dim3(256), lds, st, (const T*)x,
                                             (T*)y, mean, rstd, gamma, beta, HW, C, G, act, ppb, stat_rows));
#undef OMGSR_GN_ANY
#undef OMGSR_GN_ANY2
    return (int)hipGetLastError();
}
}  // namespace

namespace {
template <bool XF32, int YEL>
int layernorm_launch(const void* x, void* y, const float* a, const float* b, int64_t rows, int32_t C, float eps, hipStream_t st) {
    const int nch = ((C >> 3) + 63) / 64;
    auto grid = [&](int rpw) { return dim3((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))); };
    if (nch <= 1) OMGSR_DISPATCH_T(hipLaunchKernelGGL((layernorm_kernel<T, 1, 4, XF32, YEL>), grid(4), dim3(256), 0, st, x, y, a, b, rows, C, eps));
    else if (nch <= 2) OMGSR_DISPATCH_T(hipLaunchKernelGGL((layernorm_kernel<T, 2, 4, XF32, YEL>), grid(4), dim3(256), 0, st, x, y, a, b, rows, C, eps));
    else if (nch <= 3) OMGSR_DISPATCH_T(hipLaunchKernelGGL((layernorm_kernel<T, 3, 2, XF32, YEL>), grid(2), dim3(256), 0, st, x, y, a, b, rows, C, eps));
    else if (rows >= (1ll << 31)) return OMGSR_E_SHAPE;
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((layernorm_block_kernel<T, 2, XF32, YEL>), dim3((unsigned)rows), dim3(256), 0, st, x, y, a, b, C, eps));
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int omgsr_layernorm(const void* x, void* y, const float* a, const float* b, int64_t rows, int32_t C,
                               float eps, int32_t x_el, int32_t y_el, void* stream) {
    if (!x || !y || rows <= 0 || C <= 0) return OMGSR_E_BADARG;
    if ((x_el != OMGSR_EL_16 && x_el != OMGSR_EL_F32) || (y_el != OMGSR_EL_16 && y_el != OMGSR_EL_SPLIT && y_el != OMGSR_EL_MX)) return OMGSR_E_BADARG;
    if ((C & 7) || C > 64 * 8 * 8) return OMGSR_E_SHAPE;
    // the mixed-precision operand form (the consumer is an MX GEMM, igemm_gmx.hip): fp32 stream in, fp16 compute type, C % 64 == 0
    if (y_el == OMGSR_EL_MX && (x_el != OMGSR_EL_F32 || omgsr::compute_dtype() != 1 || (C & 63))) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    omgsr::TimingScope ts(OMGSR_TK_LN, 0.0, ((x_el == OMGSR_EL_F32 ? 4.0 : 2.0) + (y_el != OMGSR_EL_16 ? 4.0 : 2.0)) * (double)rows * C, st);
    if (y_el == OMGSR_EL_MX) return layernorm_launch<true, 3>(x, y, a, b, rows, C, eps, st);
    if (x_el == OMGSR_EL_F32 && y_el == OMGSR_EL_SPLIT) return layernorm_launch<true, 2>(x, y, a, b, rows, C, eps, st);
    if (x_el == OMGSR_EL_F32) return layernorm_launch<true, 0>(x, y, a, b, rows, C, eps, st);
    if (y_el == OMGSR_EL_SPLIT) return layernorm_launch<false, 2>(x, y, a, b, rows, C, eps, st);
    return layernorm_launch<false, 0>(x, y, a, b, rows, C, eps, st);
}

extern "C" int omgsr_softmax_rows(const float* s, void* p, int64_t rows, int32_t L, int32_t Lvalid, void* stream) {
    if (!s || !p || rows <= 0 || L <= 0 || Lvalid <= 0 || Lvalid > L) return OMGSR_E_BADARG;
    if ((L & 3) || L > 256 * 4 * 16 || rows >= (1ll << 31)) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    omgsr::TimingScope ts(OMGSR_TK_SOFTMAX, 0.0, 6.0 * (double)rows * L, st);
    const int nv = (L / 4 + 255) / 256;
    if (nv <= 1) OMGSR_DISPATCH_T(hipLaunchKernelGGL((softmax_rows_kernel<T, 1>), dim3((unsigned)rows), dim3(256), 0, st, s, (T*)p, L, Lvalid));
    else if (nv <= 4) OMGSR_DISPATCH_T(hipLaunchKernelGGL((softmax_rows_kernel<T, 4>), dim3((unsigned)rows), dim3(256), 0, st, s, (T*)p, L, Lvalid));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((softmax_rows_kernel<T, 16>), dim3((unsigned)rows), dim3(256), 0, st, s, (T*)p, L, Lvalid));
    return (int)hipGetLastError();
}

extern "C" int omgsr_softmax_rows_split(const float* s, void* p, int64_t rows, int32_t L, int32_t Lvalid, void* stream) {
    if (!s || !p || rows <= 0 || L <= 0 || Lvalid <= 0 || Lvalid > L) return OMGSR_E_BADARG;
    if ((L & 3) || L > 256 * 4 * 16 || rows >= (1ll << 31)) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    omgsr::TimingScope ts(OMGSR_TK_SOFTMAX, 0.0, 8.0 * (double)rows * L, st);
    const int nv = (L / 4 + 255) / 256;
    if (nv <= 1) OMGSR_DISPATCH_T(hipLaunchKernelGGL((softmax_rows_kernel<T, 1, true>), dim3((unsigned)rows), dim3(256), 0, st, s, (T*)p, L, Lvalid));
    else if (nv <= 4) OMGSR_DISPATCH_T(hipLaunchKernelGGL((softmax_rows_kernel<T, 4, true>), dim3((unsigned)rows), dim3(256), 0, st, s, (T*)p, L, Lvalid));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((softmax_rows_kernel<T, 16, true>), dim3((unsigned)rows), dim3(256), 0, st, s, (T*)p, L, Lvalid));
    return (int)hipGetLastError();
}

extern "C" int omgsr_rmsnorm_rope(void* x, const float* w, const float* w2, int32_t Lsplit, const float* cos_t, const float* sin_t, int32_t B, int32_t L,
                                  int32_t H, int32_t D, int64_t ld, int32_t col0, int32_t pos0, float eps, void* stream) {
    if (!x || !w || B <= 0 || L <= 0 || H <= 0) return OMGSR_E_BADARG;
    if (D != 128 || (ld & 7) || (col0 & 7) || ((cos_t == nullptr) != (sin_t == nullptr))) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * L;
    const int64_t groups = rows * H;
    omgsr::TimingScope ts(OMGSR_TK_ELT, 0.0, 4.0 * (double)groups * D, st);
    OMGSR_DISPATCH_T(hipLaunchKernelGGL(rmsnorm_rope_kernel<T>, dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, st, (T*)x, w, w2, Lsplit, cos_t,
                                        sin_t, rows, L, H, D, ld, col0, pos0, eps));
    return (int)hipGetLastError();
}
