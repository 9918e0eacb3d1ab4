// Shared epilogue of the implicit-GEMM kernels: 32x32 MFMA fragments -> per-wave fp32 LDS region ->
// 8 consecutive output channels per lane -> bias / activation / gate / residual -> 16-byte stores.
#pragma once
#include "common.hip.h"
#include "../../include/omgsr_hip.h"

// epi: this wave's private LDS region of 32 x (WTN + 4) floats. Fragment row-block i covers output rows
// mb[i] .. mb[i] + nvalid[i] - 1 (nvalid <= 32: rows past the tensor / past the image edge are dropped);
// n_base: first packed column of the wave tile. Must be called by every wave of the block (barriers).
template <int WTN, int FM, int FN>
OMGSR_DEVINL void igemm_epilogue(const omgsr_igemm_args& p, f32x16_t (&acc)[FM][FN], float* epi, const int lane,
                                 const int (&mb)[FM], const int (&nvalid)[FM], const int n_base, const int bz) {
    constexpr int EPI_LD = WTN + 4;
    const bool geglu = (p.act == OMGSR_ACT_GEGLU);
    const int cols_per_row = geglu ? WTN / 2 : WTN;     // produced output columns per staged row
    const int lanes_per_row = cols_per_row / 8;
    const int rows_per_pass = 64 / lanes_per_row;
    const int lrow = lane / lanes_per_row, lcol = (lane % lanes_per_row) * 8;
    bf16_t* outb = (bf16_t*)p.out + (int64_t)bz * p.out_bstride;
    float* outf = (float*)p.out + (int64_t)bz * p.out_bstride;
    const bf16_t* resb = p.residual ? (const bf16_t*)p.residual + (int64_t)bz * p.out_bstride : nullptr;
    const bool vec_ok = (p.Cout & 7) == 0 && (p.out_ld & 7) == 0;
    const int64_t ldo = p.out_ld > 0 ? p.out_ld : p.Cout;

#pragma unroll
    for (int i = 0; i < FM; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                epi[cfrag_row(lane, r) * EPI_LD + j * 32 + (lane & 31)] = acc[i][j][r];
        __syncthreads();
        for (int rb = 0; rb < 32; rb += rows_per_pass) {
            const int row = rb + lrow;
            const int m = mb[i] + row;
            float v[8];
            int n;  // first logical output column of this lane
            if (geglu) {
                // staged columns: per 64-wide group [32 a | 32 g]
                const int grp = lcol >> 5, within = lcol & 31;
                const float* pa = epi + row * EPI_LD + grp * 64 + within;
                const int nb = n_base + grp * 64 + within;   // packed bias index of a
                n = (n_base >> 1) + lcol;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float a = pa[e] * p.alpha, gt = pa[32 + e] * p.alpha;
                    if (p.bias) { a += p.bias[nb + e]; gt += p.bias[nb + 32 + e]; }
                    v[e] = a * gelu_erf_f(gt);
                }
            } else {
                const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(epi + row * EPI_LD + lcol);
                const f32x4_t x1 = *reinterpret_cast<const f32x4_t*>(epi + row * EPI_LD + lcol + 4);
                n = n_base + lcol;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = x0[e] * p.alpha; v[4 + e] = x1[e] * p.alpha; }
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (n + e < p.Cout) v[e] += p.bias[n + e];
                }
                if (p.act == OMGSR_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                } else if (p.act == OMGSR_ACT_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                }
            }
            if (row >= nvalid[i] || n >= p.Cout) continue;
            if (p.gate) {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (n + e < p.Cout) v[e] *= p.gate[n + e];
            }
            if (p.out_layout == OMGSR_LAYOUT_NHWC) {
                const int64_t o = (int64_t)m * ldo + n;
                const int64_t ro = (int64_t)m * p.Cout + n;
                if (vec_ok) {
                    if (resb) {
                        float rf[8];
                        unpack8(*reinterpret_cast<const u32x4_t*>(resb + ro), rf);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rf[e];
                    }
                    if (p.out_dtype == OMGSR_OUT_BF16) {
                        *reinterpret_cast<u32x4_t*>(outb + o) = pack8(v);
                    } else {
                        *reinterpret_cast<f32x4_t*>(outf + o) = (f32x4_t){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4_t*>(outf + o + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
                    }
                } else {
                    for (int e = 0; e < 8 && n + e < p.Cout; ++e) {
                        float x = v[e];
                        if (resb) x += (float)resb[ro + e];
                        if (p.out_dtype == OMGSR_OUT_BF16) outb[o + e] = (bf16_t)x; else outf[o + e] = x;
                    }
                }
            } else {  // OMGSR_LAYOUT_T: out[(m / t_rows) * Cout + n][m % t_rows]
                const int blk = m / p.t_rows, mr = m - blk * p.t_rows;
                for (int e = 0; e < 8 && n + e < p.Cout; ++e) {
                    const int64_t o = ((int64_t)blk * p.Cout + n + e) * p.t_ld + mr;
                    if (p.out_dtype == OMGSR_OUT_BF16) outb[o] = (bf16_t)v[e]; else outf[o] = v[e];
                }
            }
        }
    }
}

// linear-M helper for the GEMM-shaped kernels: row block i starts at m_base + 32*i
template <int WTN, int FM, int FN>
OMGSR_DEVINL void igemm_epilogue_linear(const omgsr_igemm_args& p, const int M, f32x16_t (&acc)[FM][FN], float* epi,
                                        const int lane, const int m_base, const int n_base, const int bz) {
    int mb[FM], nv[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        mb[i] = m_base + 32 * i;
        const int left = M - mb[i];
        nv[i] = left < 0 ? 0 : (left > 32 ? 32 : left);
    }
    igemm_epilogue<WTN, FM, FN>(p, acc, epi, lane, mb, nv, n_base, bz);
}

struct IgemmGeo {
    int tiles_x, tiles_y;   // halo kernel: spatial tile grid per image
    int M;          // N*Ho*Wo rows per batch entry
    int HoWo;
    int Hv, Wv;     // virtual (post-upsample) input extent
    int nk;         // K steps
    int ntm, ntn;   // tile counts
};
