// Implicit-GEMM convolution / linear / batched GEMM for gfx950 (MI355X), bf16 in, fp32 accumulate.
//
// Replaces the cuDNN/cuBLAS calls under diffusers' ResnetBlock2D / Upsample2D / Downsample2D /
// Attention / FeedForward on the OMGSR hot path.
//
// Structure (CDNA4-first, not a port of a CUDA tiling):
//   * activations NHWC bf16, weights KRSC bf16  -> both MFMA operands are contiguous along the
//     contraction index, so every fragment is one 16-byte ds_read_b128
//   * v_mfma_f32_32x32x16_bf16; 4 waves (one per SIMD) per workgroup; block tile BM x BN, BK = 32
//   * register-staged double buffering (issue global loads for K-step t+1, run the MFMAs of step t,
//     then write the staged registers to the other LDS buffer): one barrier per K-step
//   * LDS rows padded to 80 B = 5 x 16-B slots (odd) => the 16-lane service groups of ds_read_b128
//     touch 16 distinct slots: conflict-free
//   * im2col is never materialised: per K-step a thread recomputes (tap, channel) incrementally and
//     gathers its 16-byte chunk, zero-filling the padding halo; nearest-2x upsampling is folded
//     into the gather (vy >> 1)
//   * epilogue goes through LDS in fp32 so global stores/residual loads are 16 B per lane
//   * blockIdx -> tile mapping is XCD-aware (tiles sharing input rows stay on one XCD's L2)
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "timing.hip.h"
#include "igemm_epilogue.hip.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

namespace omgsr {
int igemm_dma_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st);
int igemm_halo_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st, bool phase = false);
int igemm_halo_launch_multi(const omgsr_igemm_args* a, const IgemmGeo* g0, int count, hipStream_t st, bool phase);
int igemm_halo_launch_splitk(const omgsr_igemm_args& a, const IgemmGeo& g0, int splits, hipStream_t st);     // igemm_halo_multi.hip
bool igemm_p8_wanted(const omgsr_igemm_args& a, const IgemmGeo& g);
int igemm_p8_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st);
bool igemm_gmx_ok(const omgsr_igemm_args& a);
int igemm_gmx_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st);
int igemm_halo_tiles(const omgsr_igemm_args& a, bool phase = false);
bool igemm_halo_out6_ok(const omgsr_igemm_args& a);   // igemm_halo_out6.hip: the instantiations whose epilogue writes OMGSR_EL_MX6 can run this problem
int igemm_halo_flat(const omgsr_igemm_args& a);      // pitch of the FLAT form the halo kernel would use for this problem (narrow maps), 0 = spatial tiles
int igemm_halo_tiles_form(const omgsr_igemm_args& a, int flat);
int igemm_halo_gn_slots(const omgsr_igemm_args& a, bool phase = false);
bool igemm_halo_gn_geometry_ok(const omgsr_igemm_args& a);     // igemm_halo_gn.hip: what the normalising patch producer can run
}

namespace omgsr { static int g_batch_invariant = 0; }
extern "C" int omgsr_set_batch_invariant(int on) { omgsr::g_batch_invariant = (on != 0); return 0; }

namespace {

// The argument block the POLICY looks at: in batch-invariant mode one sample's worth of rows (see omgsr_set_batch_invariant)
omgsr_igemm_args policy_view(const omgsr_igemm_args& a) {
    omgsr_igemm_args p = a;
    if (omgsr::g_batch_invariant && a.batch == 1) {
        const int64_t howo = (int64_t)a.Ho * a.Wo;
        const int64_t rows = a.sample_rows > 0 ? a.sample_rows : howo;
        if (rows >= howo) p.N = (int32_t)(rows / howo);             // conv: one image
        else { p.N = 1; p.Ho = 1; p.Wo = (int32_t)rows; }            // token matrix [1, 1, M, K]: one sequence of `rows` tokens
        if (p.N < 1) p.N = 1;
    }
    return p;
}

constexpr int BK = 32;        // K elements per pipeline step
constexpr int ROWB = 80;      // LDS row pitch in bytes (64 B of data + 16 B pad)
constexpr int NTHREADS = 256;

using Geo = IgemmGeo;

template <typename T, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(const omgsr_igemm_args p, const Geo g) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;   // wave tile
    constexpr int FM = WTM / 32, FN = WTN / 32;     // 32x32 fragments per wave
    constexpr int A_CH = BM / 64, B_CH = BN / 64;   // 16-byte chunks staged per thread
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(FM >= 1 && FN >= 1 && A_CH >= 1, "tile");
    constexpr int B_CHN = (B_CH >= 1) ? B_CH : 1;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int EPI_BYTES = 4 * 32 * (WTN + 4) * 4;
    constexpr int LDS_BYTES = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    const int tile = xcd_remap(blockIdx.x, g.ntm * g.ntn);
    const int tn = tile % g.ntn, tm = tile / g.ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = blockIdx.z;

    const T* __restrict__ in = (const T*)p.in + (int64_t)bz * p.in_bstride;
    const T* __restrict__ wt = (const T*)p.weight + (int64_t)bz * p.w_bstride;

    // ---- per-thread staging coordinates -------------------------------------------------
    const int kc = t & 3;      // which 16-byte chunk of the 64-byte K-step row
    const int r0 = t >> 2;     // 0..63
    int a_img[A_CH], a_vy0[A_CH], a_vx0[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int m = m0 + r0 + 64 * i;
        if (m < g.M) {
            const int img = m / g.HoWo;
            const int rem = m - img * g.HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_img[i] = img;
            a_vy0[i] = oy * p.stride - p.pad_top;
            a_vx0[i] = ox * p.stride - p.pad_left;
        } else {
            a_img[i] = -1; a_vy0[i] = 0; a_vx0[i] = 0;
        }
    }
    // chunk cursor: this thread's chunk at K-step kt covers k = kt*32 + kc*8 .. +8 = (tap, c0..c0+8)
    int c0 = kc * 8, tap_r = 0, tap_s = 0;
    while (c0 >= p.Cin) { c0 -= p.Cin; if (++tap_s == p.S) { tap_s = 0; ++tap_r; } }
    const int ild = p.in_ld > 0 ? p.in_ld : p.Cin;     // physical channels per pixel row; contraction channels >= ild wrap (w_lo segment)

    u32x4_t a_reg[A_CH], b_reg[B_CHN];

    auto load_stage = [&](int kt) {
        const bool tap_ok = tap_r < p.R;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int vy = a_vy0[i] + tap_r, vx = a_vx0[i] + tap_s;
            const bool ok = tap_ok && a_img[i] >= 0 && (unsigned)vy < (unsigned)g.Hv && (unsigned)vx < (unsigned)g.Wv;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (ok) {
                const int iy = vy >> p.upsample, ix = vx >> p.upsample;
                const int64_t pix = ((int64_t)a_img[i] * p.H + iy) * p.W + ix;
                v = *reinterpret_cast<const u32x4_t*>(in + pix * ild + (c0 >= ild ? c0 - ild : c0));
            }
            a_reg[i] = v;
        }
        if constexpr (B_CH >= 1) {
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                const int n = n0 + r0 + 64 * i;
                b_reg[i] = *reinterpret_cast<const u32x4_t*>(wt + (int64_t)n * p.K_pad + kt * BK + kc * 8);
            }
        } else {  // BN == 32: only threads with r0 < 32 carry a weight chunk
            if (r0 < BN) b_reg[0] = *reinterpret_cast<const u32x4_t*>(wt + (int64_t)(n0 + r0) * p.K_pad + kt * BK + kc * 8);
        }
        // advance the (tap, channel) cursor by one K-step (32 channels)
        c0 += BK;
        while (c0 >= p.Cin) { c0 -= p.Cin; if (++tap_s == p.S) { tap_s = 0; ++tap_r; } }
    };
    auto write_stage = [&](int buf) {
        unsigned char* As = lds + buf * STAGE_BYTES;
        unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_CH; ++i)
            *reinterpret_cast<u32x4_t*>(As + (r0 + 64 * i) * ROWB + kc * 16) = a_reg[i];
        if constexpr (B_CH >= 1) {
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                *reinterpret_cast<u32x4_t*>(Bs + (r0 + 64 * i) * ROWB + kc * 16) = b_reg[i];
        } else {
            if (r0 < BN) *reinterpret_cast<u32x4_t*>(Bs + r0 * ROWB + kc * 16) = b_reg[0];
        }
    };

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    load_stage(0);
    write_stage(0);
    __syncthreads();

    const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;
    for (int kt = 0; kt < g.nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < g.nk) load_stage(kt + 1);
        const unsigned char* As = lds + buf * STAGE_BYTES + wm * WTM * ROWB + frag_off;
        const unsigned char* Bs = lds + buf * STAGE_BYTES + BM * ROWB + wn * WTN * ROWB + frag_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x8_t<T> af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const x8_t<T>*>(As + i * 32 * ROWB + ks * 32);
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const x8_t<T>*>(Bs + j * 32 * ROWB + ks * 32);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mfma32(bf[j], af[i], acc[i][j]);   // transposed tile
        }
        if (kt + 1 < g.nk) write_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: fragments -> LDS (fp32) -> 8 consecutive channels per lane -------------
    float* epi = reinterpret_cast<float*>(lds) + wave * 32 * (WTN + 4);
    igemm_epilogue_linear<T, WTN, FM, FN>(p, g.M, acc, epi, lane, m0 + wm * WTM, n0 + wn * WTN, bz, p.gn_partial ? g.HoWo : 0);
}

template <int BM, int BN, int WGM, int WGN>
int launch(const omgsr_igemm_args& a, Geo g, hipStream_t st) {
    g.ntm = (g.M + BM - 1) / BM;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    g.ntn = (logical_cols + BN - 1) / BN;   // Cout_pad (multiple of 128) always covers ntn * BN rows
    dim3 grid(g.ntm * g.ntn, 1, a.batch);
    OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WGM, WGN>), grid, dim3(NTHREADS), 0, st, a, g));
    return (int)hipGetLastError();
}

// split-K reduce: out[m][n] = epilogue(sum_s partial[s][m][n]); one thread per 8 output channels.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const omgsr_igemm_args p, const float* __restrict__ ws, const int M,
                                                            const int ldw, const int splits) {
    const bool geglu = (p.act == OMGSR_ACT_GEGLU);
    const int oct_per_row = (p.Cout + 7) >> 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * oct_per_row) return;
    const int m = (int)(i / oct_per_row), n = (int)(i - (int64_t)m * oct_per_row) * 8;
    float v[8];
    if (geglu) {
        // packed columns: output n..n+7 <- a at (n/32)*64 + n%32, gate 32 further
        const int pc = (n >> 5) * 64 + (n & 31);
        float a[8], gt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = 0.0f; gt[e] = 0.0f; }
        for (int sidx = 0; sidx < splits; ++sidx) {
            const float* r = ws + ((int64_t)sidx * M + m) * ldw + pc;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] += r[e]; gt[e] += r[32 + e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = a[e] * p.alpha, y = gt[e] * p.alpha;
            if (p.bias) { x += p.bias[pc + e]; y += p.bias[pc + 32 + e]; }
            v[e] = x * gelu_erf_f(y);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        for (int sidx = 0; sidx < splits; ++sidx) {          // (ldw % 128 == 0, n % 8 == 0: two aligned 16-byte loads per partial row)
            const f32x4_t* r = reinterpret_cast<const f32x4_t*>(ws + ((int64_t)sidx * M + m) * ldw + n);
            const f32x4_t r0 = r[0], r1 = r[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = v[e] * p.alpha;
            if (p.bias && n + e < p.Cout) x += p.bias[n + e];
            if (p.act == OMGSR_ACT_SILU) x = silu_f(x);
            else if (p.act == OMGSR_ACT_GELU_TANH) x = gelu_tanh_f(x);
            v[e] = x;
        }
    }
    const int64_t ldo = p.out_ld > 0 ? p.out_ld : p.Cout;
    float amax = 0.0f;          // fp16 range guard, as in the fused epilogues: largest magnitude written as a 16-bit value
    if constexpr (std::is_same<T, f16_t>::value) {
        if (p.out_mx) {         // the mixed-precision operand form of the next GEMM (Cout % 64 == 0: whole octets), as the fused epilogues write it
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = v[e];
                if (p.gate) x *= p.gate[n + e];
                if (p.residual) x += p.res_el == OMGSR_EL_F32 ? ((const float*)p.residual)[(int64_t)m * p.Cout + n + e]
                                                               : (float)((const T*)p.residual)[(int64_t)m * p.Cout + n + e];
                v[e] = x;
                amax = fmaxf(amax, fabsf(x));
            }
            if (p.out_mx == 6) store8_mx6<T>(p.out, (int64_t)m * 4 * p.Cout, p.Cout, n, v);        // (thread per octet, whole rows: lane quads = blocks)
            else store8_mx<T>(p.out, (int64_t)m * 4 * p.Cout, p.Cout, n, v);
            if (p.overflow_flag && amax > 65504.0f) atomicOr(p.overflow_flag, 1u);
            if (p.overflow_flag && p.out_mx == 1 && amax > 448.0f) atomicOr(p.overflow_flag, 2u);
            return;
        }
    }
    if ((p.Cout & 7) == 0 && (ldo & 7) == 0 && (p.out_lo_off & 7) == 0) {      // whole octets: 16-byte residual loads and stores (round 5: 80 launches per one-image call)
        if (p.gate) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.gate[n + e];
        }
        if (p.residual) {
            float rf[8];
            if (p.res_el == OMGSR_EL_F32) load8<T, true>(p.residual, (int64_t)m * p.Cout + n, rf);
            else load8<T, false>(p.residual, (int64_t)m * p.Cout + n, rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rf[e];
        }
        const int64_t o = (int64_t)m * ldo + n;
        if (p.out_dtype == OMGSR_OUT_BF16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
            if (p.out_lo_off > 0) store8<T, 2>(p.out, o, p.out_lo_off, v);
            else store8<T, 0>(p.out, o, 0, v);
            if constexpr (std::is_same<T, f16_t>::value) {
                if (p.overflow_flag && amax > 65504.0f) atomicOr(p.overflow_flag, 1u);
            }
        } else store8<T, 1>(p.out, o, 0, v);
        return;
    }
    for (int e = 0; e < 8 && n + e < p.Cout; ++e) {
        float x = v[e];
        if (p.gate) x *= p.gate[n + e];
        if (p.residual) x += p.res_el == OMGSR_EL_F32 ? ((const float*)p.residual)[(int64_t)m * p.Cout + n + e]
                                                       : (float)((const T*)p.residual)[(int64_t)m * p.Cout + n + e];
        if (p.out_dtype == OMGSR_OUT_BF16) {
            // same overflow behaviour as the fused epilogues (pack2 / split8): fp16 saturates at +-65504 instead of hi = inf, lo = -inf
            amax = fmaxf(amax, fabsf(x));
            if constexpr (std::is_same<T, f16_t>::value) x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
            const T hi = (T)x;
            ((T*)p.out)[(int64_t)m * ldo + n + e] = hi;
            if (p.out_lo_off > 0) ((T*)p.out)[(int64_t)m * ldo + p.out_lo_off + n + e] = (T)(x - (float)hi);
        } else ((float*)p.out)[(int64_t)m * ldo + n + e] = x;
    }
    if constexpr (std::is_same<T, f16_t>::value) {
        if (p.overflow_flag && p.out_dtype == OMGSR_OUT_BF16 && amax > 65504.0f) atomicOr(p.overflow_flag, 1u);
    }
}

bool use_halo(const omgsr_igemm_args& a);
bool use_halo_phase(const omgsr_igemm_args& a);
bool out_mx6_ok(const omgsr_igemm_args& a);
int halo_splitk_plan(const omgsr_igemm_args& a);

// Split-K for the halo-tile kernel (round 5: the reference's own operating point is ONE 128 -> 512 image per call, infer/infer_omgsr_s.py:92 - its 3x3
// convs are 16 ... 64 workgroup tiles on 512 slots, 180 ... 1080 K-steps each). The contraction is cut into up to 8 ranges of 32-channel chunks; every range
// runs as one member of a igemm_halo_multi_kernel launch group (same weights, its own chunk range in IgemmGeo.cc0 / cc1, an fp32 partial tile as
// output) and splitk_reduce_kernel folds the partials into the real epilogue. A mixed-precision problem is cut on both sides of its fp16 / fp8 boundary
// (equal K-step counts: the fp8 chunks cover twice the channels), never across it. `a` is the policy view. OMGSR_HALO_SPLITK=0 for A/B.
int halo_splitk_plan(const omgsr_igemm_args& a) {
    static const char* off = getenv("OMGSR_HALO_SPLITK");
    if (off && off[0] == '0') return 1;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    if (a.in_ld != 0 || a.gn_scale_shift || a.out_layout != OMGSR_LAYOUT_NHWC || a.act == OMGSR_ACT_GEGLU || logical_cols < 96 || a.upsample ||
        a.group_tiles != 0 || a.batch != 1) return 1;
    // the mixed-precision problems on the spatial form: the only convs the halo kernel takes whatever their tile count (a plain conv with too few
    // tiles goes to the LDS-DMA kernel, which has its own split-K); one extra instantiation instead of a run-time range in every one of them
    if (a.mx_chunks16 <= 0 || omgsr::igemm_halo_flat(a) != 0) return 1;
    const int tiles = omgsr::igemm_halo_tiles(a);
    static const char* mt = getenv("OMGSR_HALO_SPLITK_MAX_TILES");
    static const int max_tiles = mt ? atoi(mt) : 128;                  // one quarter of the 512 workgroup slots
    if (tiles <= 0 || tiles >= max_tiles) return 1;
    const int nk = a.Cin / 32;
    const bool mx = a.mx_chunks16 > 0;
    static const char* tg = getenv("OMGSR_HALO_SPLITK_TARGET");
    static const int target = tg ? atoi(tg) : 512;                     // aim at ~512 workgroups: two per CU (256 ... 768 measured within 1 %)
    int splits = target / tiles;
    if (splits > 8) splits = 8;
    if (mx) {
        splits &= ~1;
        const int n16 = a.mx_chunks16, n8 = nk - n16;
        while (splits >= 2 && (n16 / (splits / 2) < 2 || n8 / (splits / 2) < 2)) splits -= 2;
    } else {
        while (splits >= 2 && nk / splits < 2) --splits;
    }
    return splits < 2 ? 1 : splits;
}

// Split-K policy: small-M problems whose 256x128 tiles cannot fill the 256 CUs but whose contraction is long.
int splitk_plan(const omgsr_igemm_args& a_real, int64_t M64) {
    const omgsr_igemm_args a = policy_view(a_real);
    if (omgsr::g_batch_invariant) M64 = (int64_t)a.N * a.Ho * a.Wo;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    if (use_halo_phase(a)) return 1;
    if (use_halo(a)) return halo_splitk_plan(a);      // the halo-tile kernel takes the problem: whole (it then owns the fused GroupNorm statistics) or, when its tiles
                                                      // cannot fill the chip, as up to 8 chunk ranges of the contraction (round 5)
    if (a.batch != 1 || a.out_layout != OMGSR_LAYOUT_NHWC || (a.Cin % 32) || (a.in_ld % 32) || logical_cols < 96 || a.mx_chunks16 > 0) return 1;
    if (a.out_mx && a.act == OMGSR_ACT_GEGLU) return 1;               // (the reduce pass writes the MX form for plain columns only)
    const int nk = a.K_pad / 32;
    const int64_t tiles = ((M64 + 255) / 256) * ((logical_cols + 127) / 128);
    const int slots = 512;                                             // two 256 x 128 workgroups per CU
    if (tiles >= slots / 2 || nk < 48) return 1;
    int splits = (int)(slots / tiles);
    if (splits > 8) splits = 8;
    while (splits > 1 && nk / splits < 16) --splits;
    return splits < 2 ? 1 : splits;
}

// A mixed-precision (MX) problem: fp16 chunks followed by block-scaled fp8 chunks. Only the halo-tile kernel's wide nine-tap shape runs it.
bool mx_geometry_ok(const omgsr_igemm_args& a) {
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    const bool common = a.R == 3 && a.S == 3 && a.stride == 1 && a.pad_top == 1 && a.pad_left == 1 && (a.Cin % 128) == 0 && a.in_ld == 0 && a.batch == 1 &&
                        logical_cols >= 96 && a.act != OMGSR_ACT_GEGLU && a.out_layout == OMGSR_LAYOUT_NHWC && a.mx_chunks16 == a.Cin / 64 &&
                        omgsr::compute_dtype() == 1;
    if (a.upsample) return common && a.weight_ph && a.Ho == 2 * a.H && a.Wo == 2 * a.W && (a.Cout & 7) == 0 && !a.residual;      // phase form
    return common && a.weight_cm && a.Ho == a.H && a.Wo == a.W;
}

// the halo-tile kernel's preconditions + the "enough tiles to fill the chip" policy
bool use_halo(const omgsr_igemm_args& a_real) {
    if (a_real.mx_chunks16 > 0) return !a_real.upsample && mx_geometry_ok(a_real);   // no other kernel understands the format
    const omgsr_igemm_args a = policy_view(a_real);              // geometry tests below do not involve N; the tile count does
    static const char* mode = getenv("OMGSR_IGEMM_MODE");
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    const bool halo_ok = a.weight_cm && a.R == 3 && a.S == 3 && a.stride == 1 && a.pad_top == 1 && a.pad_left == 1 &&
                         (a.Cin % 32) == 0 && (a.in_ld % 32) == 0 && a.batch == 1 && a.Ho == (a.H << a.upsample) && a.Wo == (a.W << a.upsample) && a.Wo >= 16 &&
                         (logical_cols >= 96 || (logical_cols <= 32 && a.act != OMGSR_ACT_GEGLU)) &&
                         a.out_layout == OMGSR_LAYOUT_NHWC;
    if (!halo_ok || (mode && (!strcmp(mode, "reg") || !strcmp(mode, "dma")))) return false;
    if (mode && !strcmp(mode, "halo")) return true;
    // the halo tile is 32 pixels wide: on narrow maps (the UNet's 16 x 16 level) half of every tile would be padding
    const int padded_w = ((a.Wo + 31) / 32) * 32;
    if (padded_w * 3 > a.Wo * 4 && !omgsr::igemm_halo_flat(a)) return false;      // > 1/3 of the columns wasted (the FLAT form wastes 2 of W + 2)
    const int tiles = omgsr::igemm_halo_tiles(a);
    const int gt = a.group_tiles < 0 ? -a.group_tiles : a.group_tiles;       // group_tiles: the launch group this problem belongs to (omgsr_igemm_multi_plan; < 0: on the FLAT form)
    return (tiles > gt ? tiles : gt) >= 192;
}

// out_mx = 6: the problem must run on one of the halo-tile kernel's OUT6 instantiations (the spatial nine-tap form, plain / split fp16 or fp6 operand);
// with a workspace and a split-K plan the reduce pass writes the form instead (its partial launches are ordinary fp32 ones)
bool out_mx6_ok(const omgsr_igemm_args& a) {
    if (a.upsample || a.act == OMGSR_ACT_GEGLU || a.out_dtype != OMGSR_OUT_BF16 || a.out_layout != OMGSR_LAYOUT_NHWC) return false;
    return omgsr::igemm_halo_out6_ok(a) && use_halo(a);
}

// Nearest-2x upsampling + 3x3 conv as four 2 x 2 convolutions of the low-res map (weight_ph: phase-summed kernels), 4 / 9 of the MFMA work
bool use_halo_phase(const omgsr_igemm_args& a_real) {
    if (a_real.mx_chunks16 > 0) return a_real.upsample && mx_geometry_ok(a_real);
    const omgsr_igemm_args a = policy_view(a_real);
    static const char* off = getenv("OMGSR_UPSAMPLE_PHASES");        // A/B runs: "0" = the gather form (nine taps on the virtual map)
    static const char* mode = getenv("OMGSR_IGEMM_MODE");
    if ((off && off[0] == '0') || (mode && (!strcmp(mode, "reg") || !strcmp(mode, "dma")))) return false;
    const bool ok = a.weight_ph && a.upsample == 1 && a.R == 3 && a.S == 3 && a.stride == 1 && a.pad_top == 1 && a.pad_left == 1 &&
                    (a.Cin % 32) == 0 && (a.in_ld % 32) == 0 && a.batch == 1 && a.Ho == 2 * a.H && a.Wo == 2 * a.W && a.W >= 16 &&
                    a.Cout >= 96 && (a.Cout & 7) == 0 && a.act != OMGSR_ACT_GEGLU && a.out_layout == OMGSR_LAYOUT_NHWC && !a.residual;
    if (!ok) return false;
    const int padded_w = ((a.W + 31) / 32) * 32;
    if (padded_w * 3 > a.W * 4) return false;
    const int tiles = omgsr::igemm_halo_tiles(a, true);
    const int gt = a.group_tiles < 0 ? -a.group_tiles : a.group_tiles;
    return (tiles > gt ? tiles : gt) >= 192;
}

}  // namespace

namespace {
// Where and in which form omgsr_igemm can emit the GroupNorm statistics of its output (0 slots = it cannot).
void gn_plan(const omgsr_igemm_args& a, int* nslot, int* entries) {
    *nslot = 0; *entries = 0;
    if (a.gn_groups <= 0 || a.Cout <= 0 || (a.Cout % a.gn_groups)) return;
    const int gsz = a.Cout / a.gn_groups;
    const bool pow2 = gsz == 4 || gsz == 8 || gsz == 16 || gsz == 32 || gsz == 64;
    const int64_t ldo = a.out_ld > 0 ? a.out_ld : a.Cout;
    if ((a.Cout & 7) || (ldo & 7) || a.act == OMGSR_ACT_GEGLU || a.out_layout != OMGSR_LAYOUT_NHWC || a.batch != 1) return;   // the epilogue's 16-byte-row fast path
    const int64_t M64 = (int64_t)a.N * a.Ho * a.Wo;
    if (use_halo_phase(a)) {
        *nslot = omgsr::igemm_halo_gn_slots(a, true);            // one slot per (wave tile, phase)
        *entries = pow2 ? a.gn_groups : a.Cout;
        return;
    }
    if (use_halo(a)) {
        if (a.Cout < 96) return;                                 // the narrow halo shape does not emit statistics
        *nslot = omgsr::igemm_halo_gn_slots(a);                  // one slot per wave tile (it lies in one image)
        *entries = pow2 ? a.gn_groups : a.Cout;
        return;
    }
    if (a.workspace && splitk_plan(a, M64) > 1) return;          // the split-K reduce pass does not emit them
    // batch-invariant mode: the GEMM-shaped kernels fold a 32-row block's statistics in an order that depends on their wave tile
    // (which the dispatcher picks from the TOTAL row count); the stand-alone statistics pass has one order per image
    if (omgsr::g_batch_invariant) return;
    const int howo = a.Ho * a.Wo;
    if (howo % 32) return;                                       // GEMM-shaped kernels: one slot per 32-row block, never straddling images
    *nslot = howo / 32;
    *entries = (pow2 && gsz <= 32) ? a.gn_groups : a.Cout;       // a 32-column wave tile holds whole groups up to 32 channels
}
}  // namespace

namespace {
// GroupNorm apply as the conv's patch producer (omgsr_igemm_args.gn_scale_shift): the problem must take the halo-tile kernel's spatial nine-tap
// form on its own merits (tile count / launch-group policy included) - the fused form is an instantiation of THAT kernel, not a fall-back
// path of its own. OMGSR_GN_FUSE=0 switches it off (A/B runs: every caller then runs the apply pass).
bool gn_fusable(const omgsr_igemm_args& a_in) {
    static const char* off = getenv("OMGSR_GN_FUSE");
    if (off && off[0] == '0') return false;
    omgsr_igemm_args a = a_in;
    a.gn_scale_shift = nullptr;                                  // the decision never depends on whether the table is attached yet
    if (!omgsr::igemm_halo_gn_geometry_ok(a) || a.out_layout != OMGSR_LAYOUT_NHWC || a.act == OMGSR_ACT_GEGLU) return false;
    if ((int64_t)a.H * a.W * a.Cin >= (1ll << 31)) return false;
    if (a.workspace && splitk_plan(a, (int64_t)a.N * a.Ho * a.Wo) > 1) return false;
    if (!use_halo(a) || use_halo_phase(a)) return false;
    if (omgsr::igemm_halo_flat(a) != 0) return false;            // (narrow maps take the FLAT form, which has no normalising instantiation)
    // Policy (one-box A/B, profiles/r05_experiments.md): the producer's VALU work costs every workgroup tile ~+22 % of its time, whatever the
    // layer; what it saves is the apply pass over the INPUT, which every 128-column output tile of a pixel block repeats. One column tile
    // (Cout <= 128: the full-resolution levels, where the apply pass is largest): -0.17 ms per layer; two: break-even; four: a loss.
    static const char* mc = getenv("OMGSR_GN_FUSE_MAX_COUT");   // A/B runs
    static const int max_cout = mc ? atoi(mc) : 128;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    return logical_cols <= max_cout;
}
}  // namespace

extern "C" int32_t omgsr_igemm_gn_fusable(const omgsr_igemm_args* ap) { return (ap && gn_fusable(*ap)) ? 1 : 0; }
extern "C" int32_t omgsr_igemm_out_mx6_ok(const omgsr_igemm_args* ap) {
    if (!ap) return 0;
    omgsr_igemm_args a = *ap;
    a.out_mx = 6;
    return out_mx6_ok(a) ? 1 : 0;
}

extern "C" int32_t omgsr_igemm_gn_slots(const omgsr_igemm_args* ap) {
    if (!ap) return 0;
    int nslot, entries;
    gn_plan(*ap, &nslot, &entries);
    return nslot;
}

extern "C" int32_t omgsr_igemm_gn_entries(const omgsr_igemm_args* ap) {
    if (!ap) return 0;
    int nslot, entries;
    gn_plan(*ap, &nslot, &entries);
    return entries;
}

extern "C" int64_t omgsr_igemm_workspace_bytes(const omgsr_igemm_args* ap) {
    if (!ap) return 0;
    const int64_t M64 = (int64_t)ap->N * ap->Ho * ap->Wo;
    const int splits = splitk_plan(*ap, M64);
    if (splits < 2) return 0;
    const int logical_cols = (ap->act == OMGSR_ACT_GEGLU) ? 2 * ap->Cout : ap->Cout;
    return (int64_t)splits * M64 * (((logical_cols + 127) / 128) * 128) * 4;
}

namespace {
// argument checks shared by omgsr_igemm and omgsr_igemm_multi; normalises in_ld (in_ld == Cin -> 0)
int validate_args(omgsr_igemm_args& a) {
    if (!a.in || !a.weight || !a.out) return OMGSR_E_BADARG;
    if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.Cin <= 0 || a.Cout <= 0 || a.Ho <= 0 || a.Wo <= 0 ||
        a.R <= 0 || a.S <= 0 || a.stride <= 0 || a.batch <= 0) return OMGSR_E_BADARG;
    if ((a.Cin & 7) || (a.Cout_pad & 127) || (a.K_pad % BK) || a.K_pad < a.R * a.S * a.Cin) return OMGSR_E_SHAPE;
    if (a.upsample != 0 && a.upsample != 1) return OMGSR_E_BADARG;
    const int64_t M64 = (int64_t)a.N * a.Ho * a.Wo;
    if (M64 >= (1ll << 31)) return OMGSR_E_SHAPE;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    if (logical_cols > a.Cout_pad) return OMGSR_E_SHAPE;
    if (a.act == OMGSR_ACT_GEGLU && ((a.Cout & 31) || a.out_layout != OMGSR_LAYOUT_NHWC)) return OMGSR_E_SHAPE;
    if (a.out_layout == OMGSR_LAYOUT_T && (a.t_rows <= 0 || a.t_ld < a.t_rows || a.residual)) return OMGSR_E_BADARG;
    if (a.out_ld != 0 && (a.out_ld < a.Cout || a.out_layout != OMGSR_LAYOUT_NHWC)) return OMGSR_E_BADARG;
    if (a.res_el != OMGSR_EL_16 && a.res_el != OMGSR_EL_F32) return OMGSR_E_BADARG;
    if (a.res_el == OMGSR_EL_F32 && a.residual && a.act == OMGSR_ACT_GEGLU) return OMGSR_E_SHAPE;
    if (a.out_lo_off < 0) return OMGSR_E_BADARG;
    if (a.out_lo_off && (a.out_dtype != OMGSR_OUT_BF16 || a.out_layout != OMGSR_LAYOUT_NHWC || (a.Cout & 7) || (a.out_ld & 7) || (a.out_lo_off & 7) ||
                         a.out_lo_off < a.Cout || a.out_ld < a.out_lo_off + a.Cout || a.gn_partial)) return OMGSR_E_SHAPE;
    if (a.in_split && ((a.in_ld > 0 ? a.in_ld : a.Cin) & 15)) return OMGSR_E_SHAPE;
    if (a.in_ld < 0 || (a.in_ld & 7) || (a.in_ld > 0 && (a.in_ld > a.Cin || a.Cin > 2 * a.in_ld))) return OMGSR_E_SHAPE;
    if (a.in_ld == a.Cin) a.in_ld = 0;
    const int ksegs = 1 + (a.in_split ? 1 : 0) + (a.w_split ? 1 : 0);      // K-concat segments of one logical channel set
    if (a.Cin % (8 * ksegs)) return OMGSR_E_SHAPE;
    // an MX operand is understood by the halo-tile kernel (3x3 convs, mx_geometry_ok) and by the MX GEMM kernel (1x1: igemm_gmx.hip)
    if (a.mx_chunks16 < 0 || (a.mx_chunks16 > 0 && !mx_geometry_ok(a) && !omgsr::igemm_gmx_ok(a))) return OMGSR_E_SHAPE;
    // ... the fp6 form (OMGSR_EL_MX6) by the halo-tile kernel's nine-tap forms only
    if (a.mx_chunks16 > 0 && a.mx_fmt != 0 && a.mx_fmt != 8 && (a.mx_fmt != 6 || !mx_geometry_ok(a))) return OMGSR_E_SHAPE;
    // out_mx = 6 (the fp6 operand form as OUTPUT): the halo-tile kernel's dedicated instantiations (igemm_halo_out6.hip) and the split-K reduce pass -
    // omgsr_igemm_out_mx6_ok() tells the host beforehand; everything else writes a stream tensor and the cast kernel follows
    if (a.out_mx != 0 && a.out_mx != 1 && a.out_mx != 6) return OMGSR_E_BADARG;
    if (a.out_mx == 6 && !out_mx6_ok(a)) return OMGSR_E_SHAPE;
    if (a.out_mx && (a.out_dtype != OMGSR_OUT_BF16 || a.out_layout != OMGSR_LAYOUT_NHWC || (a.Cout & 63) || a.out_lo_off || a.out_ld || a.gn_partial ||
                     omgsr::compute_dtype() != 1)) return OMGSR_E_SHAPE;
    if (a.gn_scale_shift && (a.gn_nimg <= 0 || a.gn_act != OMGSR_ACT_SILU || !gn_fusable(a))) return OMGSR_E_SHAPE;      // (SiLU is the one activation the producer applies)
    if (a.gn_partial) {                            // must be exactly what omgsr_igemm_gn_slots / _gn_entries promised
        int nslot, entries;
        gn_plan(a, &nslot, &entries);
        if (nslot <= 0 || a.gn_entries != entries) return OMGSR_E_BADARG;
    }
    return 0;
}

void work_of(const omgsr_igemm_args& a, double* flops, double* bytes) {
    const int64_t M64 = (int64_t)a.N * a.Ho * a.Wo;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    const int ksegs = 1 + (a.in_split ? 1 : 0) + (a.w_split ? 1 : 0);
    // algorithmic work: a split operand's / weight's extra K segments are precision overhead, not useful FLOPs
    *flops = 2.0 * (double)M64 * (double)a.R * a.S * (a.Cin / ksegs) * (double)logical_cols * a.batch;
    const double out_b = (a.out_dtype == OMGSR_OUT_F32 ? 4.0 : (a.out_lo_off ? 4.0 : 2.0)), res_b = a.residual ? (a.res_el == OMGSR_EL_F32 ? 4.0 : 2.0) : 0.0;
    *bytes = (2.0 * ((double)a.N * a.H * a.W * (a.in_ld > 0 ? a.in_ld : a.Cin) + (double)a.Cout_pad * a.K_pad) + (double)M64 * a.Cout * (out_b + res_b)) * a.batch;
}

Geo geo_of(const omgsr_igemm_args& a) {
    Geo g{};
    g.M = (int)((int64_t)a.N * a.Ho * a.Wo);
    g.HoWo = a.Ho * a.Wo;
    g.Hv = a.H << a.upsample;
    g.Wv = a.W << a.upsample;
    g.nk = a.K_pad / BK;
    g.splits = 1; g.nk_total = g.nk;
    return g;
}
}  // namespace

// Problems of ONE layer that differ only in their tensors and spatial extents (the tiled VAE's tile-shape groups): plan first
// (omgsr_igemm_multi_plan writes each problem's `group_tiles` so that kernel choice and GroupNorm-statistics layout are decided for
// the GROUP), then launch. Problems that take the halo-tile kernel in the same shape run as one launch; the rest one by one.
extern "C" int omgsr_igemm_multi_plan(omgsr_igemm_args* args, int32_t count) {
    if (!args || count <= 0) return OMGSR_E_BADARG;
    static const char* off = getenv("OMGSR_MULTI");           // A/B runs: "0" = every problem its own launch
    int total = 0, total_ph = 0, total_flat = 0;
    bool all_flat_ok = true;
    for (int i = 0; i < count; ++i) {
        args[i].group_tiles = 0;
        total += omgsr::igemm_halo_tiles_form(args[i], 0);
        total_ph += omgsr::igemm_halo_tiles(args[i], true);
        const int tf = omgsr::igemm_halo_tiles_form(args[i], 1);
        all_flat_ok = all_flat_ok && tf > 0;
        total_flat += tf > 0 ? tf : 0;
    }
    if ((off && off[0] == '0') || omgsr::g_batch_invariant) return 0;       // batch-invariant mode: every decision from one sample alone
    // one form per group: FLAT (group_tiles < 0) when every problem can run it and the group then needs fewer workgroup tiles
    const bool flat = all_flat_ok && total_flat < total;
    for (int i = 0; i < count; ++i) args[i].group_tiles = (args[i].upsample && args[i].weight_ph) ? total_ph : (flat ? -total_flat : total);
    return 0;
}

extern "C" int omgsr_igemm_multi(const omgsr_igemm_args* args, int32_t count, void* stream) {
    if (!args || count <= 0) return OMGSR_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    constexpr int MAXG = 8;
    omgsr_igemm_args grp[MAXG];
    Geo geo[MAXG];
    int ng = 0, mode = -1;          // mode of the open group: 0 = halo 9 taps, 1 = phase form
    double gf = 0.0, gb = 0.0;
    long long gm = 0;
    auto flush = [&]() -> int {
        if (ng == 0) return 0;
        int rc;
        {
            const int logical_cols = (grp[0].act == OMGSR_ACT_GEGLU) ? 2 * grp[0].Cout : grp[0].Cout;
            omgsr::TimingScope ts(OMGSR_TK_IGEMM, gf, gb, st, gm, logical_cols, (long long)grp[0].R * grp[0].S * grp[0].Cin);
            ts.rec.variant = ng == 1 ? (mode == 1 ? 6 : 3) : (mode == 1 ? 8 : 7);      // 7 / 8: igemm_halo_multi_kernel (gather / phase form)
            if (grp[0].gn_scale_shift) ts.rec.variant = ng == 1 ? 10 : 11;              // 10 / 11: the GroupNorm-fused instantiations
            if (grp[0].mx_chunks16 > 0 && grp[0].mx_fmt == 6) ts.rec.variant = mode == 1 ? (ng == 1 ? 16 : 17) : (ng == 1 ? 13 : 14);      // 13 / 14 (/ 15: split-K): fp6 correction chunks; 16 / 17: in the phase form
            rc = ng == 1 ? omgsr::igemm_halo_launch(grp[0], geo[0], st, mode == 1) : omgsr::igemm_halo_launch_multi(grp, geo, ng, st, mode == 1);
        }
        ng = 0; gf = gb = 0.0; gm = 0;
        return rc;
    };
    for (int i = 0; i < count; ++i) {
        omgsr_igemm_args a = args[i];
        int rc = validate_args(a);
        if (rc != 0) { flush(); return rc; }
        const int m = (a.workspace && splitk_plan(a, (int64_t)a.N * a.Ho * a.Wo) > 1) ? -1 : use_halo_phase(a) ? 1 : use_halo(a) ? 0 : -1;
        const bool same = ng > 0 && m == mode && a.weight == grp[0].weight && a.Cin == grp[0].Cin && a.in_ld == grp[0].in_ld && a.Cout == grp[0].Cout &&
                          a.act == grp[0].act && a.out_dtype == grp[0].out_dtype && a.out_lo_off == grp[0].out_lo_off && a.out_ld == grp[0].out_ld &&
                          a.res_el == grp[0].res_el && (a.residual != nullptr) == (grp[0].residual != nullptr) &&
                          (a.gn_partial != nullptr) == (grp[0].gn_partial != nullptr) && a.gn_entries == grp[0].gn_entries && a.bias == grp[0].bias &&
                          a.gate == grp[0].gate && a.alpha == grp[0].alpha && a.upsample == grp[0].upsample && a.mx_chunks16 == grp[0].mx_chunks16 && a.mx_fmt == grp[0].mx_fmt && a.out_mx == grp[0].out_mx &&
                          ((a.Cout <= 32) == (grp[0].Cout <= 32)) && ((omgsr::igemm_halo_flat(a) != 0) == (omgsr::igemm_halo_flat(grp[0]) != 0)) &&
                          (a.gn_scale_shift != nullptr) == (grp[0].gn_scale_shift != nullptr) && a.gn_act == grp[0].gn_act;
        if (m < 0) {                     // not a halo problem: its own launch, in order
            rc = flush();
            if (rc == 0) rc = omgsr_igemm(&args[i], stream);
            if (rc != 0) return rc;
            continue;
        }
        if (ng > 0 && (!same || ng == MAXG)) { rc = flush(); if (rc != 0) return rc; }
        mode = m;
        double f, b;
        work_of(a, &f, &b);
        gf += f; gb += b; gm += (long long)a.N * a.Ho * a.Wo;
        geo[ng] = geo_of(a);
        grp[ng++] = a;
    }
    return flush();
}

extern "C" int omgsr_igemm(const omgsr_igemm_args* ap, void* stream) {
    if (!ap) return OMGSR_E_BADARG;
    omgsr_igemm_args a = *ap;
    {
        const int rc = validate_args(a);
        if (rc != 0) return rc;
    }
    const int64_t M64 = (int64_t)a.N * a.Ho * a.Wo;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    Geo g{};
    g.M = (int)M64;
    g.HoWo = a.Ho * a.Wo;
    g.Hv = a.H << a.upsample;
    g.Wv = a.W << a.upsample;
    g.nk = a.K_pad / BK;
    hipStream_t st = (hipStream_t)stream;
    double flops, bytes;
    work_of(a, &flops, &bytes);
    omgsr::TimingScope ts(OMGSR_TK_IGEMM, flops, bytes, st, M64 * a.batch, logical_cols, (long long)a.R * a.S * a.Cin);
    // Tile choice: the 128x128 tile is the MFMA-efficient default; narrow outputs use 128x32 so
    // padded columns do not burn MFMA cycles; small problems drop to 64x64 to fill the 256 CUs.
    // batch-invariant mode: every kernel-family decision below looks at ONE sample's rows (policy_view), like use_halo / splitk_plan
    const omgsr_igemm_args pv = policy_view(a);
    const int64_t Mp = omgsr::g_batch_invariant ? (int64_t)pv.N * pv.Ho * pv.Wo : M64;
    const int64_t pbatch = omgsr::g_batch_invariant ? 1 : a.batch;
    const int64_t tiles128 = ((Mp + 127) / 128) * (a.Cout_pad / 128) * pbatch;
    // Large problems: LDS-DMA kernel (256x128 tile, 3-stage ring). OMGSR_IGEMM_MODE=reg|dma overrides (A/B runs).
    static const char* mode = getenv("OMGSR_IGEMM_MODE");
    const int64_t tiles256 = ((Mp + 255) / 256) * ((logical_cols + 127) / 128) * pbatch;
    g.splits = 1; g.nk_total = g.nk;
    if (a.workspace && !(mode && (!strcmp(mode, "reg") || !strcmp(mode, "halo")))) {
        const int splits = splitk_plan(a, M64);
        if (splits > 1) {
            if (a.gn_partial) return OMGSR_E_BADARG;        // the reduce pass does not emit GroupNorm statistics (omgsr_igemm_gn_slots says so)
            int rc;
            if (use_halo(a)) {                       // chunk ranges of the halo-tile kernel as one launch group
                ts.rec.variant = a.mx_fmt == 6 ? 15 : 12;
                rc = omgsr::igemm_halo_launch_splitk(a, g, splits, st);
            } else {
                g.splits = splits;
                ts.rec.variant = 4;
                rc = omgsr::igemm_dma_launch(a, g, st);
            }
            if (rc != 0) return rc;
            const int ldw = ((logical_cols + 127) / 128) * 128;
            const int64_t items = M64 * ((a.Cout + 7) / 8);
            OMGSR_DISPATCH_T(hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, a,
                                                (const float*)a.workspace, (int)M64, ldw, splits));
            return (int)hipGetLastError();
        }
    }
    // GEMM-shaped problem over a mixed-precision operand (the UNet's transformer-block linears in the accurate tier)
    if (a.mx_chunks16 > 0 && omgsr::igemm_gmx_ok(a)) {
        // wide projections (GEGLU, q|k, N >= 1280): the ping-pong kernel's 256 x 256 tile moves 2/3 of the operand bytes of 256 x 128 through the
        // LDS-DMA path, which is what bounds these short-K problems (profiles/r04_experiments.md). OMGSR_P8_MX=0 for A/B.
        static const char* pm = getenv("OMGSR_P8_MX");
        if (!(pm && pm[0] == '0') && !omgsr::g_batch_invariant && a.batch == 1 && omgsr::igemm_p8_wanted(a, g)) {
            ts.rec.variant = 5;
            return omgsr::igemm_p8_launch(a, g, st);
        }
        ts.rec.variant = 9;
        return omgsr::igemm_gmx_launch(a, g, st);
    }
    // 3x3 s1 p1 convs with a chunk-major weight copy: halo-tile kernel (input patch reused by all 9 taps)
    if (use_halo_phase(a)) { ts.rec.variant = (a.mx_chunks16 > 0 && a.mx_fmt == 6) ? 16 : 6; return omgsr::igemm_halo_launch(a, g, st, true); }
    if (use_halo(a)) { ts.rec.variant = a.gn_scale_shift ? 10 : (a.mx_chunks16 > 0 && a.mx_fmt == 6) ? 13 : 3; return omgsr::igemm_halo_launch(a, g, st); }
    if (a.gn_scale_shift) return OMGSR_E_SHAPE;                  // (validate_args already refused it: never reached)
    {
        static const char* dbg = getenv("OMGSR_DEBUG_DISPATCH");      // one line per conv-shaped problem that did NOT take the halo kernel
        if (dbg && a.R == 3)
            fprintf(stderr, "[omgsr] 3x3 off the halo path: N %d H %d W %d Cin %d Cout %d stride %d pad %d,%d ups %d Ho %d Wo %d cm %d batch %d act %d layout %d tiles %d\n",
                    a.N, a.H, a.W, a.Cin, a.Cout, a.stride, a.pad_top, a.pad_left, a.upsample, a.Ho, a.Wo, a.weight_cm != nullptr, a.batch, a.act, a.out_layout,
                    omgsr::igemm_halo_tiles(a));
    }
    const bool dma_ok = logical_cols >= 96 && (a.Cin % 32) == 0 && (a.in_ld % 32) == 0;   // the DMA kernel's K-steps never straddle taps (or the wrap point)
    // ... and (round 5) GEMM-shaped problems too small to fill the chip with 256-row tiles: the LDS-DMA kernel's 64 / 128-row instantiations
    // (the register-staged kernel's 64 x 64 tile spends ~0.6 us per 32-channel K-step on them). OMGSR_DMA_SMALL=0 for A/B.
    static const char* dsm = getenv("OMGSR_DMA_SMALL");
    const bool gemm_shaped = a.R == 1 && a.S == 1 && a.stride == 1 && a.pad_top == 0 && a.pad_left == 0 && !a.upsample && a.Ho == a.H && a.Wo == a.W;
    const bool dma_small = !(dsm && dsm[0] == '0') && gemm_shaped && a.batch == 1 && a.K_pad >= 256 && a.out_layout == OMGSR_LAYOUT_NHWC && Mp >= 64;
    if (dma_ok && ((mode && !strcmp(mode, "dma")) || (!(mode && !strcmp(mode, "reg")) && (tiles256 >= 192 || dma_small)))) {
        static const char* p8 = getenv("OMGSR_P8");               // A/B runs: "0" = never use the ping-pong GEMM kernel
        if (!(p8 && p8[0] == '0') && !omgsr::g_batch_invariant && omgsr::igemm_p8_wanted(a, g)) {
            ts.rec.variant = 5;
            return omgsr::igemm_p8_launch(a, g, st);
        }
        ts.rec.variant = 2;
        return omgsr::igemm_dma_launch(a, g, st);
    }
    ts.rec.variant = 1;
    if (a.act == OMGSR_ACT_GEGLU) return launch<128, 128, 2, 2>(a, g, st);   // needs a 64-wide wave tile
    if (logical_cols <= 32) return launch<128, 32, 4, 1>(a, g, st);
    if (logical_cols <= 64 || tiles128 < 192) return launch<64, 64, 2, 2>(a, g, st);
    return launch<128, 128, 2, 2>(a, g, st);
}
