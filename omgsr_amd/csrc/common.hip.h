// Shared device helpers for the OMGSR gfx950 kernels.
// Wave = 64 lanes; MFMA fragment layouts follow the CDNA4 ISA (v_mfma_f32_32x32x16_bf16):
//   A frag (32 x 16): lane l holds row (l & 31), k = 8*(l >> 5) + j, j = 0..7
//   B frag (16 x 32): lane l holds col (l & 31), k = 8*(l >> 5) + j
//   C/D   (32 x 32): lane l holds col (l & 31), row = (r & 3) + 8*(r >> 2) + 4*(l >> 5), r = 0..15
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// The 16-bit element type T of activations / weights is a template parameter of every kernel:
// bf16 (the reference's default --weight_dtype) or fp16 (its other 16-bit option: same MFMA rate, 10-bit
// mantissa). The library-wide choice is omgsr_set_compute_dtype(); fp32 accumulation either way.
typedef _Float16 f16_t;
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <typename T> struct vec8;
template <> struct vec8<bf16_t> { typedef bf16x8_t type; };
template <> struct vec8<f16_t> { typedef f16x8_t type; };
template <typename T> using x8_t = typename vec8<T>::type;
namespace omgsr { int compute_dtype(); }     // 0 = bf16, 1 = fp16 (elementwise.hip)
#define OMGSR_DISPATCH_T(...)                                               \
    do {                                                                    \
        if (omgsr::compute_dtype() == 1) { using T = f16_t; __VA_ARGS__; }  \
        else { using T = bf16_t; __VA_ARGS__; }                             \
    } while (0)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));

#define OMGSR_DEVINL __device__ __forceinline__

OMGSR_DEVINL f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
OMGSR_DEVINL f32x16_t mfma32(f16x8_t a, f16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// row index inside a 32x32 C fragment held by (lane, reg)
OMGSR_DEVINL int cfrag_row(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

OMGSR_DEVINL float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }

template <typename T> OMGSR_DEVINL void unpack8(const u32x4_t v, float (&f)[8]);
template <> OMGSR_DEVINL void unpack8<bf16_t>(const u32x4_t v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
template <> OMGSR_DEVINL void unpack8<f16_t>(const u32x4_t v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned int w = v[i];
        const f16x2_t h = *reinterpret_cast<const f16x2_t*>(&w);
        f[2 * i] = (float)h[0];
        f[2 * i + 1] = (float)h[1];
    }
}

template <typename T> OMGSR_DEVINL unsigned int pack2(float lo, float hi);
template <> OMGSR_DEVINL unsigned int pack2<bf16_t>(float lo, float hi) {
    bf16x2_t r = __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t);
    return *reinterpret_cast<unsigned int*>(&r);
}
template <> OMGSR_DEVINL unsigned int pack2<f16_t>(float lo, float hi) {
    // saturate instead of overflowing to inf (what diffusers' fp16 paths do with clip(-65504, 65504))
    lo = __builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f);
    hi = __builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f);
    f16x2_t r = __builtin_convertvector((f32x2_t){lo, hi}, f16x2_t);
    return *reinterpret_cast<unsigned int*>(&r);
}

template <typename T> OMGSR_DEVINL u32x4_t pack8(const float (&f)[8]) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2<T>(f[2 * i], f[2 * i + 1]);
    return v;
}

// ---- element-kind generic 8-channel access (OMGSR_EL_*: include/omgsr_hip.h) --------------------------------
// load: 8 consecutive channels as floats from a 16-bit (one 16-byte load) or an fp32 (two 16-byte loads) row
template <typename T, bool F32> OMGSR_DEVINL void load8(const void* base, const int64_t idx, float (&f)[8]) {
    if constexpr (F32) {
        const float* p = reinterpret_cast<const float*>(base) + idx;
        const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(p), hi = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    } else {
        unpack8<T>(*reinterpret_cast<const u32x4_t*>(reinterpret_cast<const T*>(base) + idx), f);
    }
}
// the two-term split of 8 values: hi = round(v), lo = round(v - hi), both in the compute type
template <typename T> OMGSR_DEVINL float lo_of(unsigned int h2, int which);
template <> OMGSR_DEVINL float lo_of<bf16_t>(unsigned int h2, int which) { return which ? __uint_as_float(h2 & 0xffff0000u) : __uint_as_float(h2 << 16); }
template <> OMGSR_DEVINL float lo_of<f16_t>(unsigned int h2, int which) {
    const f16x2_t h = *reinterpret_cast<const f16x2_t*>(&h2);
    return (float)h[which];
}
template <typename T> OMGSR_DEVINL void split8(const float (&v)[8], u32x4_t& hi, u32x4_t& lo) {
    // pair by pair (two live temporaries instead of sixteen: the igemm epilogue calls this with ~250 registers in use)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned int h2 = pack2<T>(v[2 * i], v[2 * i + 1]);
        hi[i] = h2;
        lo[i] = pack2<T>(v[2 * i] - lo_of<T>(h2, 0), v[2 * i + 1] - lo_of<T>(h2, 1));
    }
}
// ---- the MX operand form (OMGSR_EL_MX, fp16 compute type): a row of C logical channels is 4C bytes,
//   [a_hi fp16 (2C B) | a_lo' fp8 e4m3 (C B) | a_hi' fp8 e4m3 (C B)],   a_lo' = (a - a_hi) * 2^OMGSR_MX_LO_SHIFT,  a_hi' = a_hi
// The hi half meets the fp16 weights in fp16 MFMAs; the two fp8 thirds meet fp8 copies of w_hi / w_lo in block-scaled MFMAs
// (v_mfma_scale_f32_32x32x64_f8f6f4: 64 channels per instruction at twice the fp16 rate, the E8M0 scale operands put the 2^-11 back),
// all into one fp32 accumulator: a_hi w_hi + a_lo w_hi + a_hi w_lo with the two correction terms carried to 2^-4 of THEIR size
// (2^-16 of the product) for one third less MFMA time and operand traffic than three fp16 segments.
// v_cvt_pk_fp8_f32 (gfx950: OCP e4m3fn, round to nearest even) turns |x| >= 464 into NaN: clamp to +-448 first.
#define OMGSR_MX_LO_SHIFT 11
OMGSR_DEVINL unsigned int pack4_fp8(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f); b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f); d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned int)v;
}
// 8 channels c .. c + 7 of the row that starts at byte `row_byte0` and holds C logical channels
template <typename T> OMGSR_DEVINL void store8_mx(void* base, const int64_t row_byte0, const int C, const int c, const float (&f)[8]) {
    unsigned char* row = reinterpret_cast<unsigned char*>(base) + row_byte0;
    u32x4_t hi;
    u32x2_t lo8, hi8;
    // four values at a time (eight live temporaries: the igemm epilogue calls this with ~250 registers in use)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned int h0 = pack2<T>(f[4 * q], f[4 * q + 1]), h1 = pack2<T>(f[4 * q + 2], f[4 * q + 3]);
        hi[2 * q] = h0; hi[2 * q + 1] = h1;
        const float a0 = lo_of<T>(h0, 0), a1 = lo_of<T>(h0, 1), a2 = lo_of<T>(h1, 0), a3 = lo_of<T>(h1, 1);
        const float s = (float)(1 << OMGSR_MX_LO_SHIFT);
        lo8[q] = pack4_fp8((f[4 * q] - a0) * s, (f[4 * q + 1] - a1) * s, (f[4 * q + 2] - a2) * s, (f[4 * q + 3] - a3) * s);
        hi8[q] = pack4_fp8(a0, a1, a2, a3);
    }
    *reinterpret_cast<u32x4_t*>(row + 2 * c) = hi;
    *reinterpret_cast<u32x2_t*>(row + 2 * C + c) = lo8;
    *reinterpret_cast<u32x2_t*>(row + 3 * C + c) = hi8;
}

// ---- the fp6 operand form (OMGSR_EL_MX6, round 5; include/omgsr_hip.h): the same 4C-byte row whose two correction thirds hold e2m3 codes with
// one E8M0 scale byte per 32-channel block. A 64-byte group of a third = 64 channels = two blocks; block h owns bytes [16h, 16h + 16) and
// [32 + 16h, 40 + 16h) (its 192-bit code string, channel i at bits [6i, 6i + 6)), byte 40 + 16h (scale) and 7 zero bytes - exactly what lane-half h of
// the halo-tile kernel's fragment reads fetches. No 2^11 on the low part: the block scale carries every magnitude.
template <int CTRL> OMGSR_DEVINL unsigned quad_perm(const unsigned v) {          // DPP quad_perm: 0xB1 = lanes ^ 1, 0x4E = lanes ^ 2
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
OMGSR_DEVINL float quad_max(float m) {
    m = fmaxf(m, __uint_as_float(quad_perm<0xB1>(__float_as_uint(m))));
    return fmaxf(m, __uint_as_float(quad_perm<0x4E>(__float_as_uint(m))));
}
// E8M0 byte of a block whose largest magnitude is m: the largest scaled value lands in e2m3's top binade [4, 8) (7.5 < x < 8 saturates)
OMGSR_DEVINL unsigned mx6_scale_byte(const float m) {
    const int eb = (int)(__float_as_uint(m) >> 23) - 2;
    return (unsigned)(eb < 0 ? 0 : eb);
}
// e2m3 code (sign | 2 exponent bits | 3 mantissa bits) of s = value / block scale, round to nearest even, saturating at +-7.5:
// the grid is k/8 below 2, k/4 below 4, k/2 up to 7.5, so code = rint(|s| / step) + 8 * (binade index above [1, 2))
OMGSR_DEVINL unsigned e2m3_code(const float s) {
    const float a = fminf(fabsf(s), 7.5f);
    unsigned eb = __float_as_uint(a) >> 23;
    eb = eb < 127u ? 127u : eb;
    const float inv_step = __uint_as_float((257u - eb) << 23);          // 8, 4, 2 for |s| in [0, 2), [2, 4), [4, 7.5]
    const unsigned q = (unsigned)(int)__builtin_rintf(a * inv_step);
    return (q + ((eb - 127u) << 3)) | ((__float_as_uint(s) >> 26) & 32u);
}
// eight values -> 48 bits (w0: 32, w1: 16)
OMGSR_DEVINL void e2m3_pack8(const float (&v)[8], const float inv_scale, unsigned& w0, unsigned& w1) {
    unsigned c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = e2m3_code(v[e] * inv_scale);
    w0 = c[0] | (c[1] << 6) | (c[2] << 12) | (c[3] << 18) | (c[4] << 24) | (c[5] << 30);
    w1 = (c[5] >> 2) | (c[6] << 4) | (c[7] << 10);
}
// 8 channels c .. c + 7 of a row. COOPERATIVE: lanes 4k .. 4k + 3 of the wave must hold the four octets (c & 31) = 0, 8, 16, 24 of ONE block of ONE
// row and all of them must CALL this (the thread-per-octet kernels - cast, GroupNorm apply, split-K reduce - and the GEMM epilogues map
// consecutive lanes to consecutive octets and C % 64 == 0). `store` false (a row outside the image, the same for the whole quad): nothing is written.
template <typename T> OMGSR_DEVINL void store8_mx6(void* base, const int64_t row_byte0, const int C, const int c, const float (&f)[8], const bool store = true) {
    unsigned char* row = reinterpret_cast<unsigned char*>(base) + row_byte0;
    u32x4_t hi;
    float ah[8], al[8], mh = 0.0f, ml = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned int h2 = pack2<T>(f[2 * q], f[2 * q + 1]);
        hi[q] = h2;
        ah[2 * q] = lo_of<T>(h2, 0); ah[2 * q + 1] = lo_of<T>(h2, 1);
        al[2 * q] = f[2 * q] - ah[2 * q]; al[2 * q + 1] = f[2 * q + 1] - ah[2 * q + 1];
        mh = fmaxf(mh, fmaxf(fabsf(ah[2 * q]), fabsf(ah[2 * q + 1])));
        ml = fmaxf(ml, fmaxf(fabsf(al[2 * q]), fabsf(al[2 * q + 1])));
    }
    if (store) *reinterpret_cast<u32x4_t*>(row + 2 * c) = hi;
    const unsigned sl = mx6_scale_byte(quad_max(ml)), sh = mx6_scale_byte(quad_max(mh));
    const int oct = (c >> 3) & 3;
    unsigned char* grp = row + 2 * C + (c >> 6) * 64 + ((c >> 5) & 1) * 16;        // the block's first 16 bytes in the a_lo' third; + C: a_hi'
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        const unsigned sb = seg ? sh : sl;
        unsigned w0, w1;
        e2m3_pack8(seg ? ah : al, __uint_as_float((254u - sb) << 23), w0, w1);
        const unsigned o0 = quad_perm<0xB1>(w0), o1 = quad_perm<0xB1>(w1);       // the odd octet's bits reach its even partner
        const unsigned d0 = w0, d1 = w1 | (o0 << 16), d2 = (o0 >> 16) | (o1 << 16);
        unsigned char* g = grp + seg * C;
        if (!store) continue;
        if (oct == 0) {
            *reinterpret_cast<u32x2_t*>(g) = u32x2_t{d0, d1};
            *reinterpret_cast<unsigned*>(g + 8) = d2;
        } else if (oct == 2) {
            *reinterpret_cast<unsigned*>(g + 12) = d0;
            *reinterpret_cast<u32x4_t*>(g + 32) = u32x4_t{d1, d2, sb, 0u};
        }
    }
}

// store: EL 0 = 16-bit at base[idx]; 1 = fp32 at base[idx]; 2 = split, hi at base[idx], lo at base[idx + lo_off]
template <typename T, int EL> OMGSR_DEVINL void store8(void* base, const int64_t idx, const int64_t lo_off, const float (&f)[8]) {
    if constexpr (EL == 1) {
        float* p = reinterpret_cast<float*>(base) + idx;
        *reinterpret_cast<f32x4_t*>(p) = (f32x4_t){f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4_t*>(p + 4) = (f32x4_t){f[4], f[5], f[6], f[7]};
    } else if constexpr (EL == 2) {
        u32x4_t hi, lo;
        split8<T>(f, hi, lo);
        *reinterpret_cast<u32x4_t*>(reinterpret_cast<T*>(base) + idx) = hi;
        *reinterpret_cast<u32x4_t*>(reinterpret_cast<T*>(base) + idx + lo_off) = lo;
    } else {
        *reinterpret_cast<u32x4_t*>(reinterpret_cast<T*>(base) + idx) = pack8<T>(f);
    }
}

// x * sigmoid(x); v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division
OMGSR_DEVINL float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact (erf) GELU. erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the 16-bit output step): one
// v_rcp + one v_exp + 7 FMAs instead of libm erff's range-split polynomials (~3x the VALU work; the GEGLU
// epilogue of the UNet's 320 -> 2x1280 projection spent a third of its time in erff)
OMGSR_DEVINL float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = 1.0f - poly * t * __expf(-z * z);        // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}
OMGSR_DEVINL float gelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    // tanh(u) = 1 - 2/(1+exp(2u)); saturates cleanly for large |u|
    float t = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
    return 0.5f * x * (1.0f + t);
}

OMGSR_DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
OMGSR_DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware bijective remap of a linear block id: blocks that land on one XCD (b % 8) get a
// contiguous range of logical tile ids so neighbouring tiles share that XCD's L2.
OMGSR_DEVINL int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Ablation switches (OMGSR_*_ABLATE: parts of a kernel compiled out, RESULTS ARE GARBAGE, timing experiments only) are honoured
// only next to OMGSR_ABLATION_OK=1, so a stray variable cannot corrupt a production run. The *_VARIANT / *_MODE / *_BM switches
// select between equivalent schedules (same results) and need no gate.
static inline const char* ablation_env(const char* name) {
    const char* ok = getenv("OMGSR_ABLATION_OK");
    return (ok && ok[0] == '1') ? getenv(name) : nullptr;
}
