// Shared epilogue of the implicit-GEMM kernels.
//
// The kernels accumulate the TRANSPOSED tile, acc = mfma(W_frag, X_frag): a 32x32 fragment then holds, per
// lane, ONE output pixel (col = lane & 31) and 16 output channels in registers, 4 consecutive channels per
// register quad (channel = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)). Each wave stages a 32-pixel row block in
// its PRIVATE fp32 LDS region with ds_write_b128 (4 channels per write; the 272-B row pitch puts the 8
// lanes of a write group on 32 distinct banks), reads it back as 8 consecutive channels per lane and
// issues 16-byte global stores: every output row leaves the CU as full 128-B segments (per-lane
// register stores at a row stride were measured 3x slower, profiles/r01_pmc_igemm.md §5).
// Only ONE block barrier (the staging region overlaps the operand ring); the rest is wave-local, LDS
// operations of one wave execute in program order.
#pragma once
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include <type_traits>

OMGSR_DEVINL void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// epi: this wave's private LDS region of 32 x (WTN + 4) floats. Fragment row-block i covers output rows
// (pixels) mb[i] .. mb[i] + nvalid[i] - 1; n_base: first PACKED column of the wave tile (GEGLU: packed
// [32 a | 32 g] per 64). Must be called by every wave of the block.
// RES32: the residual is an fp32 stream tensor (accurate tier) instead of the 16-bit compute type.
template <typename T, int WTN, int FM, int FN, bool RES32, bool OUT6 = false>
OMGSR_DEVINL void igemm_epilogue_impl(const omgsr_igemm_args& p, f32x16_t (&acc)[FM][FN], float* epi, const int lane,
                                      const int (&mb)[FM], const int (&nvalid)[FM], const int n_base, const int bz,
                                      float* gn_dst, const int gn_howo, const int pxs, const int flat_base) {
    // pxs: distance in output pixels between consecutive rows of a fragment row block (1; 2 when the block is one phase of a
    // phase-decomposed upsampling conv: its pixels land on every other column). A residual is not supported with pxs == 2.
    // pxs < 0 (halo kernel, FLAT form, fast path only): the rows of a block are consecutive positions f = mb[i] + row of the image's
    // FLATTENED PADDED map of pitch P = -pxs (f = y P + x', x' = 0 and x' = P - 1 are the zero border columns): position f is output
    // pixel flat_base + y (P - 2) + x' - 1 of the dense NHWC tensor, border positions are dropped (from the GroupNorm partials too).
    constexpr int EPI_LD = WTN + 4;
    static_assert(WTN == FN * 32, "wave tile width");
    const int half = lane >> 5, px = lane & 31;
    // a GEGLU projection never carries a residual (omgsr_igemm rejects the combination with an fp32 one): the fp32-residual
    // instantiation drops the gate path and its 8 bias registers - the halo / 256-wide kernels have none to spare
    const bool geglu = !RES32 && (p.act == OMGSR_ACT_GEGLU);
    const int cols_per_row = geglu ? WTN / 2 : WTN;     // produced output columns per staged row
    const int lanes_per_row = cols_per_row / 8;
    const int rows_per_pass = 64 / lanes_per_row;
    const int lrow = lane / lanes_per_row, lcol = (lane % lanes_per_row) * 8;
    T* outb = (T*)p.out + (int64_t)bz * p.out_bstride;
    float* outf = (float*)p.out + (int64_t)bz * p.out_bstride;
    typedef typename std::conditional<RES32, float, T>::type res_t;
    const res_t* resb = p.residual ? (const res_t*)p.residual + (int64_t)bz * p.out_bstride : nullptr;
    const int64_t ldo = p.out_ld > 0 ? p.out_ld : p.Cout;
    const bool osplit = p.out_lo_off > 0 && p.out_dtype == OMGSR_OUT_BF16;        // two-term split: lo out_lo_off columns after hi
    // fp16 range guard: the largest magnitude this lane writes as a 16-bit value; a wave that saw one past 65504 (pack2 saturates
    // there) ORs 1 into p.overflow_flag so the host can fall back instead of returning a silently clipped result
    constexpr bool OVF_CHK = std::is_same<T, f16_t>::value;
    float amax = 0.0f;
    auto note8 = [&](const float (&v)[8]) {
        if constexpr (OVF_CHK) {
            amax = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                         fmaxf(fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))), amax));
        }
    };
    auto flush_ovf = [&]() {
        if constexpr (OVF_CHK) {
            if (p.overflow_flag && p.out_dtype == OMGSR_OUT_BF16 && __any(amax > 65504.0f) && lane == 0) atomicOr(p.overflow_flag, 1u);
            // MX output: the fp8 correction fields use fixed scales and clamp at +-448 (store8_mx), so an element beyond that keeps only its
            // fp16 main term (single-rounding accuracy, nothing becomes NaN). Bit 1 of the same word counts as a DIAGNOSTIC, not an error:
            // ops.mx_saturation_seen() (real SD2.1 feed-forward / attention activations reach the hundreds; seeded weights never do)
            if (p.overflow_flag && p.out_mx && __any(amax > 448.0f) && lane == 0) atomicOr(p.overflow_flag, 2u);
        }
    };
    const bool vec_ok = (p.Cout & 7) == 0 && (ldo & 7) == 0;
    const float flat_rp = pxs < 0 ? 1.0f / (float)(-pxs) : 0.0f;
    // output pixel of row `row` of block i, and whether it is stored
    auto opix = [&](const int i, const int row, bool& ok) -> int {
        if (pxs > 0) { ok = row < nvalid[i]; return mb[i] + row * pxs; }
        const int P = -pxs, f = mb[i] + row;
        const int y = (int)(((float)f + 0.5f) * flat_rp);         // exact: f < 2^20
        const int x = f - y * P;
        ok = row < nvalid[i] && x >= 1 && x <= P - 2;
        return flat_base + f - 2 * y - 1;
    };

    // a lane's output columns are the same for every row pass: fetch bias / gate ONCE (per-pass scalar
    // loads made the epilogue latency-bound: ~30 % of a 128-channel conv's time)
    float bias_a[8], bias_g[8], gate8[8];
    {
        const int grp = lcol >> 5, within = lcol & 31;
        const int nb = geglu ? n_base + grp * 64 + within : n_base + lcol;     // packed index
        const int nlog = geglu ? (n_base >> 1) + lcol : n_base + lcol;         // logical output column
        // 8 consecutive floats per lane: two 16-byte loads each when the row of channels is whole (eight scalar
        // loads per vector made even a LayerNorm kernel 5x slower, tools/bench_gn.py)
        const bool whole = (p.Cout & 7) == 0 && (geglu || nb + 8 <= p.Cout);
        auto load8 = [&](const float* v, const int c, const float dflt, const bool on, float (&o)[8]) {
            if (on && whole) {
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(v + c), hi = *reinterpret_cast<const f32x4_t*>(v + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (on && (geglu || c + e < p.Cout)) ? v[c + e] : dflt;
            }
        };
        load8(p.bias, nb, 0.0f, p.bias != nullptr, bias_a);
        load8(p.bias, nb + 32, 0.0f, p.bias != nullptr && geglu, bias_g);
        load8(p.gate, nlog, 1.0f, p.gate != nullptr, gate8);
    }
    // fast path (NHWC, 16-byte rows): every residual load of the tile is issued up front, before the barrier and
    // the LDS staging, so the wave pays ONE global-load latency instead of one per row pass (the serial
    // load -> add -> store chain made short-K GEMMs epilogue-bound: ~25 us per 256x128 tile,
    // profiles/r01_pmc_igemm.md §7)
    constexpr int EPI_CPR = WTN;                       // staged columns per row
    const bool fast = vec_ok && p.out_layout == OMGSR_LAYOUT_NHWC;
    const int n_out = geglu ? (n_base >> 1) + lcol : n_base + lcol;      // first logical output column of this lane
    const bool col_ok = n_out < p.Cout;
    if (fast) {
        {
            constexpr int NPASS_MAX = 32 / (64 / (EPI_CPR / 8));           // non-GEGLU passes per row block
            const int npass = 32 / rows_per_pass;                          // GEGLU: half as many lanes per row -> 2x rows per pass
            // two row blocks of residual in flight (32 VGPRs): block i + 2 is requested as soon as block i is stored
            constexpr int RV = RES32 ? 2 : 1;                              // 16-byte pieces per 8 residual values
            constexpr int NB = RES32 ? 1 : 2;                              // row blocks of residual in flight: 32 VGPRs either way (the halo
                                                                           // kernel has no more to give: 238 of its 256 are live here)
            u32x4_t res[NB][NPASS_MAX][RV];
            auto load_res = [&](const int i, u32x4_t (&dst)[NPASS_MAX][RV]) {
#pragma unroll
                for (int ps = 0; ps < NPASS_MAX; ++ps) {
                    const int row = ps * rows_per_pass + lrow;
#pragma unroll
                    for (int w = 0; w < RV; ++w) dst[ps][w] = (u32x4_t){0u, 0u, 0u, 0u};
                    bool rok;
                    const int rpix = opix(i, row, rok);
                    if (ps < npass && rok && col_ok) {
                        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(resb + (int64_t)rpix * p.Cout + n_out);
#pragma unroll
                        for (int w = 0; w < RV; ++w) dst[ps][w] = src[w];
                    }
                }
            };
            if (resb) {
                load_res(0, res[0]);
                if constexpr (FM > 1 && NB > 1) load_res(1, res[NB - 1]);
            }
            // Fused GroupNorm statistics: (sum, sum of squares) of the values this wave stores, kept per channel of the
            // lane's octet while rows stream by, then folded over the rows with a fixed butterfly (deterministic) and
            // written as one entry per GROUP (group size 4 / 8 / 16 / 32 / 64) or per CHANNEL (any other group size, e.g.
            // the UNet's 10 / 20 / 40: omgsr_groupnorm_finalize folds entries-per-group). Where: one slot per wave tile
            // (gn_dst, halo kernel: the tile lies in one image) or one slot per 32-row block (gn_howo = Ho*Wo, GEMM-shaped
            // kernels with Ho*Wo % 32 == 0: a row block never straddles images).
            const bool gn_on = gn_dst != nullptr || gn_howo > 0;
            const int gsz = gn_on ? p.Cout / p.gn_groups : 8;
            const int gn_entries = p.gn_entries;                        // gn_groups (one entry per group) or Cout (per channel)
            const bool gn_grouped = gn_on && gn_entries == p.gn_groups && gsz >= 4;     // group size 1: per channel IS per group
            float gs[8], gq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gs[e] = 0.0f; gq[e] = 0.0f; }
            auto gn_flush = [&](float* dst) {
                if (gn_grouped) {
                    float a0 = (gs[0] + gs[1]) + (gs[2] + gs[3]), a1 = (gs[4] + gs[5]) + (gs[6] + gs[7]);
                    float b0 = (gq[0] + gq[1]) + (gq[2] + gq[3]), b1 = (gq[4] + gq[5]) + (gq[6] + gq[7]);
                    for (int o = lanes_per_row; o < 64; o <<= 1) {
                        a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); b0 += __shfl_xor(b0, o); b1 += __shfl_xor(b1, o);
                    }
                    if (gsz == 4) {
                        if (lrow == 0 && col_ok) *reinterpret_cast<f32x4_t*>(dst + (n_out >> 2) * 2) = (f32x4_t){a0, b0, a1, b1};
                    } else {
                        float s1 = a0 + a1, q1 = b0 + b1;
                        for (int o = 1; o * 8 < gsz; o <<= 1) { s1 += __shfl_xor(s1, o); q1 += __shfl_xor(q1, o); }
                        if (lrow == 0 && col_ok && (n_out % gsz) == 0) *reinterpret_cast<f32x2_t*>(dst + (n_out / gsz) * 2) = (f32x2_t){s1, q1};
                    }
                } else {
                    for (int o = lanes_per_row; o < 64; o <<= 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { gs[e] += __shfl_xor(gs[e], o); gq[e] += __shfl_xor(gq[e], o); }
                    }
                    if (lrow == 0 && col_ok) {
                        float* d = dst + (int64_t)n_out * 2;
#pragma unroll
                        for (int e = 0; e < 8; e += 2)
                            *reinterpret_cast<f32x4_t*>(d + 2 * e) = (f32x4_t){gs[e], gq[e], gs[e + 1], gq[e + 1]};
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { gs[e] = 0.0f; gq[e] = 0.0f; }
            };
            __syncthreads();            // every wave is done reading the operand ring the staging region overlaps
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float* wr = epi + px * EPI_LD + 4 * half;
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4_t*>(wr + j * 32 + 8 * q) =
                            (f32x4_t){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                wave_lds_fence();
#pragma unroll
                for (int ps = 0; ps < NPASS_MAX; ++ps) {
                    if (ps >= npass) break;
                    const int row = ps * rows_per_pass + lrow;
                    float v[8];
                    if (geglu) {
                        const int grp = lcol >> 5, within = lcol & 31;
                        const float* pa = epi + row * EPI_LD + grp * 64 + within;
                        const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(pa), a1 = *reinterpret_cast<const f32x4_t*>(pa + 4);
                        const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(pa + 32), g1 = *reinterpret_cast<const f32x4_t*>(pa + 36);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = (a0[e] * p.alpha + bias_a[e]) * gelu_erf_f(g0[e] * p.alpha + bias_g[e]);
                            v[4 + e] = (a1[e] * p.alpha + bias_a[4 + e]) * gelu_erf_f(g1[e] * p.alpha + bias_g[4 + e]);
                        }
                    } else {
                        const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(epi + row * EPI_LD + lcol);
                        const f32x4_t x1 = *reinterpret_cast<const f32x4_t*>(epi + row * EPI_LD + lcol + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = x0[e] * p.alpha + bias_a[e]; v[4 + e] = x1[e] * p.alpha + bias_a[4 + e]; }
                        if (p.act == OMGSR_ACT_SILU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                        } else if (p.act == OMGSR_ACT_GELU_TANH) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                        }
                    }
                    if (p.gate) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= gate8[e];
                    }
                    if (resb) {
                        float rf[8];
                        if constexpr (RES32) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { rf[e] = __uint_as_float(res[i % NB][ps][0][e]); rf[4 + e] = __uint_as_float(res[i % NB][ps][RV - 1][e]); }
                        } else {
                            unpack8<T>(res[i % NB][ps][0], rf);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rf[e];
                    }
                    bool sok;
                    const int spix = opix(i, row, sok);
                    if constexpr (OUT6) {      // OUT6 (a template flag with its own kernel instantiations, igemm_halo_out6.hip: as a run-time branch next to the
                                               // other output forms the cooperative store spilled every kernel that includes this epilogue): the output is the
                                               // OMGSR_EL_MX6 operand of an up-sampling conv. Lane quads hold the four octets of a block (consecutive lanes =
                                               // consecutive octets of one output row, Cout % 64 == 0): every lane calls, only the store is predicated
                        if (sok && col_ok) note8(v);
                        store8_mx6<T>(outb, (int64_t)spix * 4 * p.Cout, p.Cout, n_out, v, sok && col_ok);
                    } else
                    if (sok && col_ok) {
                        const int64_t o = (int64_t)spix * ldo + n_out;
                        if (gn_on) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gq[e] += v[e] * v[e]; }
                        }
                        if (p.out_mx) {          // the mixed-precision operand form of the next conv (OMGSR_EL_MX: 4 Cout bytes per pixel)
                            note8(v);
                            store8_mx<T>(outb, (int64_t)spix * 4 * p.Cout, p.Cout, n_out, v);
                        } else if (osplit) {
                            u32x4_t hi, lo;
                            note8(v);
                            split8<T>(v, hi, lo);
                            *reinterpret_cast<u32x4_t*>(outb + o) = hi;
                            *reinterpret_cast<u32x4_t*>(outb + o + p.out_lo_off) = lo;
                        } else if (p.out_dtype == OMGSR_OUT_BF16) {
                            note8(v);
                            *reinterpret_cast<u32x4_t*>(outb + o) = pack8<T>(v);
                        } else {
                            *reinterpret_cast<f32x4_t*>(outf + o) = (f32x4_t){v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4_t*>(outf + o + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
                        }
                    }
                }
                if (resb && i + NB < FM) load_res(i + NB, res[i % NB]);
                if (gn_howo > 0 && nvalid[i] > 0) {        // one slot per 32-row block (wave-uniform condition)
                    const int img = mb[i] / gn_howo;
                    const int slot = (mb[i] - img * gn_howo) >> 5;
                    gn_flush(p.gn_partial + ((int64_t)img * (gn_howo >> 5) + slot) * gn_entries * 2);
                }
                wave_lds_fence();       // this wave's reads are done before the next row block overwrites the region
            }
            if (gn_dst) gn_flush(gn_dst);
            flush_ovf();
            return;
        }
    }
    __syncthreads();            // every wave is done reading the operand ring the staging region overlaps
    // Transposed 16-bit output (V^T for the attention kernel: out[(m / t_rows) * Cout + n][m % t_rows]) with whole 32-row blocks:
    // the staged block is read DOWN its columns, 8 rows per lane, and leaves as 16-byte pieces - four lanes cover 64 contiguous
    // bytes of one output row. (The element-wise path below writes 2 bytes per store: 2x the time of the whole GEMM at K = 320,
    // +25 % on the Flux V projections.)
    if (p.out_layout == OMGSR_LAYOUT_T && p.out_dtype == OMGSR_OUT_BF16 && !geglu && (p.t_rows & 31) == 0 && (p.t_ld & 7) == 0 &&
        (WTN % 16) == 0) {
        bool whole = true;
#pragma unroll
        for (int i = 0; i < FM; ++i) whole = whole && (mb[i] & 31) == 0 && (nvalid[i] & 7) == 0;
        if (whole) {
            constexpr int NPASS_T = WTN / 16;
            const int nl = lane >> 2, g8 = (lane & 3) * 8;
            float bt[NPASS_T], gt[NPASS_T];
#pragma unroll
            for (int ps = 0; ps < NPASS_T; ++ps) {
                const int n = n_base + ps * 16 + nl;
                bt[ps] = (p.bias && n < p.Cout) ? p.bias[n] : 0.0f;
                gt[ps] = (p.gate && n < p.Cout) ? p.gate[n] : 1.0f;
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float* wr = epi + px * EPI_LD + 4 * half;
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4_t*>(wr + j * 32 + 8 * q) =
                            (f32x4_t){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                wave_lds_fence();
                const int blk = mb[i] / p.t_rows, mr0 = mb[i] - blk * p.t_rows;
#pragma unroll
                for (int ps = 0; ps < NPASS_T; ++ps) {
                    const int n = n_base + ps * 16 + nl;
                    if (n < p.Cout && g8 < nvalid[i]) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float x = epi[(g8 + e) * EPI_LD + ps * 16 + nl] * p.alpha + bt[ps];
                            if (p.act == OMGSR_ACT_SILU) x = silu_f(x);
                            else if (p.act == OMGSR_ACT_GELU_TANH) x = gelu_tanh_f(x);
                            v[e] = x * gt[ps];
                        }
                        note8(v);
                        *reinterpret_cast<u32x4_t*>(outb + ((int64_t)blk * p.Cout + n) * p.t_ld + mr0 + g8) = pack8<T>(v);
                    }
                }
                wave_lds_fence();
            }
            flush_ovf();
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        float* wr = epi + px * EPI_LD + 4 * half;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4_t*>(wr + j * 32 + 8 * q) =
                    (f32x4_t){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        wave_lds_fence();
        for (int rb = 0; rb < 32; rb += rows_per_pass) {
            const int row = rb + lrow;
            const int m = mb[i] + row * pxs;
            float v[8];
            int n;  // first logical output column of this lane
            if (geglu) {
                // staged columns: per 64-wide group [32 a | 32 g]
                const int grp = lcol >> 5, within = lcol & 31;
                const float* pa = epi + row * EPI_LD + grp * 64 + within;
                n = (n_base >> 1) + lcol;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = pa[e] * p.alpha + bias_a[e], gt = pa[32 + e] * p.alpha + bias_g[e];
                    v[e] = a * gelu_erf_f(gt);
                }
            } else {
                const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(epi + row * EPI_LD + lcol);
                const f32x4_t x1 = *reinterpret_cast<const f32x4_t*>(epi + row * EPI_LD + lcol + 4);
                n = n_base + lcol;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = x0[e] * p.alpha + bias_a[e]; v[4 + e] = x1[e] * p.alpha + bias_a[4 + e]; }
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
                for (int e = 0; e < 8; ++e) v[e] *= gate8[e];
            }
            if (p.out_layout != OMGSR_LAYOUT_NHWC) note8(v);     // (the transposed form carries no residual)
            if (p.out_layout == OMGSR_LAYOUT_NHWC) {
                const int64_t o = (int64_t)m * ldo + n;
                const int64_t ro = (int64_t)m * p.Cout + n;
                if (vec_ok) {
                    if (resb) {
                        float rf[8];
                        load8<T, RES32>(resb, ro, rf);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rf[e];
                    }
                    if (p.out_dtype == OMGSR_OUT_BF16) {
                        note8(v);               // after the residual add, like the fast path: the value that is actually stored
                        *reinterpret_cast<u32x4_t*>(outb + o) = pack8<T>(v);
                    } else {
                        *reinterpret_cast<f32x4_t*>(outf + o) = (f32x4_t){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4_t*>(outf + o + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
                    }
                } else {
                    for (int e = 0; e < 8 && n + e < p.Cout; ++e) {
                        float x = v[e];
                        if (resb) x += (float)resb[ro + e];
                        if constexpr (OVF_CHK) amax = fmaxf(amax, fabsf(x));
                        if (p.out_dtype == OMGSR_OUT_BF16) outb[o + e] = (T)x; else outf[o + e] = x;
                    }
                }
            } else {  // OMGSR_LAYOUT_T: out[(m / t_rows) * Cout + n][m % t_rows]
                const int blk = m / p.t_rows, mr = m - blk * p.t_rows;
                for (int e = 0; e < 8 && n + e < p.Cout; ++e) {
                    const int64_t o = ((int64_t)blk * p.Cout + n + e) * p.t_ld + mr;
                    if (p.out_dtype == OMGSR_OUT_BF16) outb[o] = (T)v[e]; else outf[o] = v[e];
                }
            }
        }
        wave_lds_fence();       // this wave's reads are done before the next row block overwrites the region
    }
    flush_ovf();
}

template <typename T, int WTN, int FM, int FN, bool OUT6 = false>
OMGSR_DEVINL void igemm_epilogue(const omgsr_igemm_args& p, f32x16_t (&acc)[FM][FN], float* epi, const int lane,
                                 const int (&mb)[FM], const int (&nvalid)[FM], const int n_base, const int bz,
                                 float* gn_dst = nullptr, const int gn_howo = 0, const int pxs = 1, const int flat_base = 0) {
    if (p.res_el == OMGSR_EL_F32 && p.residual) igemm_epilogue_impl<T, WTN, FM, FN, true, OUT6>(p, acc, epi, lane, mb, nvalid, n_base, bz, gn_dst, gn_howo, pxs, flat_base);
    else igemm_epilogue_impl<T, WTN, FM, FN, false, OUT6>(p, acc, epi, lane, mb, nvalid, n_base, bz, gn_dst, gn_howo, pxs, flat_base);
}

// linear-M helper for the GEMM-shaped kernels: row block i starts at m_base + 32*i
template <typename T, int WTN, int FM, int FN>
OMGSR_DEVINL void igemm_epilogue_linear(const omgsr_igemm_args& p, const int M, f32x16_t (&acc)[FM][FN], float* epi,
                                        const int lane, const int m_base, const int n_base, const int bz, const int gn_howo = 0) {
    int mb[FM], nv[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        mb[i] = m_base + 32 * i;
        const int left = M - mb[i];
        nv[i] = left < 0 ? 0 : (left > 32 ? 32 : left);
    }
    igemm_epilogue<T, WTN, FM, FN>(p, acc, epi, lane, mb, nv, n_base, bz, nullptr, gn_howo);
}

struct IgemmGeo {
    int tiles_x, tiles_y;   // halo kernel: spatial tile grid per image
    int M;          // N*Ho*Wo rows per batch entry
    int HoWo;
    int Hv, Wv;     // virtual (post-upsample) input extent
    int nk;         // K steps (per split)
    int nk_total;   // K steps of the whole contraction
    int splits;     // split-K factor (1 = none)
    int ntm, ntn;   // tile counts
    int main_prio;  // halo-tile kernel (A/B switch OMGSR_HALO_MAINPRIO): s_setprio level of a wave while it is in its K loop (0 = off); its
                    // prologue / epilogue run at priority 0, so the co-resident workgroup's MFMA stream is not delayed by epilogue VALU / LDS traffic
    int flat;       // halo-tile kernel, FLAT form (narrow maps, W <= 45): pitch P = W + 2 of the flattened padded map a tile walks (256 consecutive
                    // positions per workgroup; the nine taps are the uniform shifts dy P + dx); 0 = the 8 x 32 spatial tile
    int cc0, cc1;   // halo-tile kernel, split-K (round 5): this launch-group member runs the 32-channel chunks [cc0, cc1) of the contraction and writes
                    // an fp32 partial tile (cc1 == 0: the whole contraction). A range never straddles the fp16 / fp8 boundary of an MX problem
    int interleave; // halo-tile kernel, phase form: 1 = the four output phases of a tile are consecutive logical blocks of ONE x-only grid
                    // (they share an XCD's L2: the low-res patch is fetched from HBM once, not four times); 0 = blockIdx.y = phase
};
