/*
 * omgsr_hip.h — C ABI of libomgsr_hip.so, the gfx950 (MI355X) kernel library under the
 * diffusers-shaped module surface that OMGSR's inference pipeline consumes.
 *
 * The reference (wuer5/OMGSR) has no FFI of its own: its hot path calls diffusers modules
 * (infer/omgsr_s_infer_model.py:75-85,173; infer/omgsr_f_infer_model.py:15-18,191-211) which
 * dispatch to torch's L0 ops (F.conv2d, F.group_norm, F.linear, F.layer_norm,
 * F.scaled_dot_product_attention, F.interpolate ...).  Each entry point below replaces one such
 * L0 op family; the comment on each names the torch op / reference call
 * site it stands in for.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless stated otherwise; activations are bf16 NHWC
 *    ([N,H,W,C]; a token matrix [B,L,C] is the same thing with H*W = L)
 *  - no allocation inside the library: callers pass outputs and workspaces
 *  - `stream` is a hipStream_t passed as void* (0 = null stream)
 *  - return value: 0 on success, a hipError_t (>0) from the launch, or a negative OMGSR_E_* code
 *    for an unsupported shape/argument; nothing is launched when a negative code is returned
 */
#ifndef OMGSR_HIP_H
#define OMGSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMGSR_E_BADARG (-1)   /* null pointer / non-positive dimension            */
#define OMGSR_E_SHAPE (-2)    /* dimension not supported by the kernel (alignment) */
#define OMGSR_E_ARCH (-3)     /* device is not gfx950                               */

#define OMGSR_ACT_NONE 0
#define OMGSR_ACT_SILU 1
#define OMGSR_ACT_GELU_TANH 2
#define OMGSR_ACT_GEGLU 3     /* out[:, j] = a_j * gelu_erf(g_j); weight rows packed [32 a | 32 g] per 64 */

#define OMGSR_OUT_BF16 0
#define OMGSR_OUT_F32 1

#define OMGSR_LAYOUT_NHWC 0   /* out[m][n]                                                        */
#define OMGSR_LAYOUT_T 1      /* out[(m / t_rows) * Cout + n][m % t_rows] with row stride t_ld     */

/* Element kinds of a tensor argument (`*_el` parameters). The library's data model has two tensor classes:
 *   operand tensors  feed an MFMA: the 16-bit compute type (OMGSR_EL_16), optionally as a two-term split
 *                    x = hi + lo with both halves in the compute type (OMGSR_EL_SPLIT, outputs only): a row of C
 *                    logical channels is stored as [hi_0..hi_{C-1} | lo_0..lo_{C-1}] (row stride 2C), and the consumer's
 *                    weights are packed with their input channels duplicated, so hi*W + lo*W accumulates in fp32
 *   stream tensors   everything between two GEMMs (residual stream, conv outputs awaiting a norm, latents): the
 *                    compute type in the fast tiers, fp32 (OMGSR_EL_F32) in the accurate tier (`--weight_dtype fp32`) */
#define OMGSR_EL_16 0
#define OMGSR_EL_F32 1
#define OMGSR_EL_SPLIT 2
/* OMGSR_EL_MX (outputs of the GroupNorm apply / LayerNorm / cast kernels, omgsr_igemm's out_mx, omgsr_attention's o_mx; fp16 compute type): the three-part operand of the mixed-precision
 * split. A row of C logical channels (C % 64 == 0) is 4C bytes = 2C 16-bit slots: [a_hi fp16 | a_lo' fp8 e4m3 | a_hi' fp8 e4m3] with
 * a_lo' = (a - a_hi) * 2^11 and a_hi' = a_hi, both clamped to +-448. Its consumer (omgsr_igemm_args.mx_chunks16 > 0: the halo-tile
 * kernel for 3x3 convs, igemm_gmx_kernel for GEMM-shaped problems) multiplies the first half by fp16 weights in fp16 MFMAs and the fp8 parts by fp8 copies of w_hi / w_lo in block-scaled MFMAs
 * (v_mfma_scale_f32_32x32x64_f8f6f4, twice the fp16 rate), all into one fp32 accumulator. */
#define OMGSR_EL_MX 3
/* OMGSR_EL_MX6 (round 5; outputs of the GroupNorm apply / cast kernels and of omgsr_igemm with out_mx = 6 where omgsr_igemm_out_mx6_ok() allows): the same 4C-byte row [a_hi fp16 | a_lo' | a_hi'] whose two
 * correction thirds hold fp6 (OCP e2m3) codes with one E8M0 scale per 32-channel block: every 64-byte group of a third covers 64 channels =
 * two blocks; block h (0, 1) owns bytes [16h, 16h + 16) (codes 0 .. 20 and the low 2 bits of 21: channel i at bits [6i, 6i + 6) of the block's
 * little-endian 192-bit string), bytes [32 + 16h, 40 + 16h) (the rest of the string), byte 40 + 16h (the block's scale: a = code 2^(scale - 127))
 * and 7 zero bytes. a_lo' = a - a_hi, a_hi' = a_hi; scale = max(0, biased exponent of the block's largest magnitude - 2), codes round to
 * nearest even and saturate at +-7.5. The f8f6f4 MFMA runs such chunks in 8 passes where fp8 takes 16. */
#define OMGSR_EL_MX6 4

#define OMGSR_DT_BF16 0
#define OMGSR_DT_F16 1
/*
 * 16-bit element type of every activation / weight tensor the library touches (fp32 accumulation and
 * statistics either way). bf16 is the reference's default --weight_dtype; fp16 is its other 16-bit option
 * (infer/infer_omgsr_s.py:134-149): same MFMA rate, 10-bit mantissa (rel-L2 vs the fp32 oracle ~8x lower),
 * narrower range. Process-wide: set it before packing weights and keep it for the lifetime of those buffers.
 */
int omgsr_set_compute_dtype(int dtype);
int omgsr_get_compute_dtype(void);

/* ABI version: bump on any struct change. */
int omgsr_abi_version(void);
/* 0 when the current device is gfx950, OMGSR_E_ARCH otherwise. */
int omgsr_check_device(void);
const char* omgsr_error_string(int code);

/*
 * K1/K2/K3/K5/K6 — implicit-GEMM convolution / linear / batched GEMM on MFMA
 * (replaces F.conv2d, F.linear, F.interpolate(nearest,2x)+conv, torch.bmm/baddbmm).
 *   out[m, n] = epilogue( alpha * sum_{r,s,c} in[img, vy(r), vx(s), c] * weight[n, (r*S+s)*Cin + c] )
 * with m = (img, oy, ox); virtual input coords vy = oy*stride - pad_top + r (same for x); when
 * `upsample` is 1 the virtual input is the nearest-2x upsampling of `in` (vy>>1), never materialised.
 * epilogue(v): v += bias[n]; v = act(v); v *= gate[n]; v += residual[m, n].
 * Requirements: Cin % 8 == 0; weight packed [Cout_pad][K_pad] bf16 with Cout_pad % 128 == 0
 * (zero rows) and K_pad = roundup(R*S*Cin, 32) (zero columns).
 */
typedef struct omgsr_igemm_args {
    const void* in;        /* bf16 [batch][N,H,W,Cin]                                 */
    const void* weight;    /* bf16 [batch?][Cout_pad][K_pad]                           */
    const float* bias;     /* f32 [Cout] (GEGLU: [2*Cout] packed like the rows) | NULL */
    const float* gate;     /* f32 [Cout] | NULL                                        */
    const void* residual;  /* bf16, same layout as out (NHWC only) | NULL              */
    void* out;             /* bf16 or f32                                              */
    int32_t N, H, W, Cin;  /* input geometry (per batch entry)                         */
    int32_t Cout;          /* logical output channels (GEGLU: the halved count)        */
    int32_t Cout_pad, K_pad;
    int32_t R, S, stride, pad_top, pad_left, upsample;
    int32_t Ho, Wo;
    int32_t act, out_dtype, out_layout;
    int32_t t_rows, t_ld;  /* OMGSR_LAYOUT_T only                                      */
    int32_t out_ld;        /* NHWC row stride of `out` in elements; 0 = Cout (residual always uses Cout) */
    int32_t batch;         /* grid.z; strides below are in elements                    */
    int64_t in_bstride, w_bstride, out_bstride;
    float alpha;
    const void* weight_cm; /* optional second packing of a 3x3 weight, slice-major [Cin/32][9 taps][Cout_pad][32]
                              (Cin % 32 == 0; tap = r*3+s): enables the halo-tile kernel | NULL */
    void* workspace;       /* split-K scratch (f32), omgsr_igemm_workspace_bytes() bytes | NULL = never split */
    float* gn_partial;     /* optional fused GroupNorm statistics of `out`: f32 [N][gn_slots][gn_entries][2] (sum, sum of
                              squares per slot and entry), to be folded by omgsr_groupnorm_finalize | NULL. Only when
                              omgsr_igemm_gn_slots() > 0 for these arguments; gn_entries = omgsr_igemm_gn_entries():
                              gn_groups (one entry per group) or Cout (one per channel: group sizes that are not 4..64 pow2). */
    int32_t gn_groups;
    int32_t gn_entries;
    int32_t res_el;        /* element kind of `residual`: OMGSR_EL_16 | OMGSR_EL_F32 (fp32 residual stream) */
    int32_t in_split;      /* 0 | 1: `in` is a two-term split operand (Cin = 2 x the logical channels; the packed weight holds
                              every input channel twice). Informational: FLOP accounting, kernel selection. */
    int64_t sample_rows;   /* output rows of ONE independent sample (an image's Ho*Wo, a sequence's tokens); 0 = N*Ho*Wo. Only read in
                              batch-invariant mode (omgsr_set_batch_invariant): kernel-family / split-K decisions then depend on it
                              instead of the batch's total rows, so a batch of B gives the bits of B batch-1 calls */
    int32_t out_lo_off;    /* > 0: a 16-bit output is written as the two-term split: hi at column n, lo at column n + out_lo_off
                              of the same row (out_lo_off >= Cout, out_ld >= out_lo_off + Cout, both % 8 == 0; NHWC, Cout % 8 == 0) */
    int32_t in_ld;         /* physical channels of one input pixel row; 0 = Cin. in_ld < Cin (in_ld <= Cin <= 2 * in_ld, in_ld % 8 == 0): the
                              contraction is longer than the row and WRAPS - channel c of a tap reads input channel c - in_ld for
                              c >= in_ld. That is how a weight carried as the two-term split w = w_hi + w_lo (accurate tier, fp32
                              checkpoints that are not 16-bit representable) meets the operand without copying it: the packed weight
                              holds [w_hi | w_lo] per tap against an operand [a] (Cin = 2C, in_ld = C), or [w_hi | w_hi | w_lo] against a
                              split operand [a_hi | a_lo] (Cin = 3C, in_ld = 2C: a_hi w_hi + a_lo w_hi + a_hi w_lo in one fp32 accumulator) */
    int32_t w_split;       /* 0 | 1: the packed weight carries the extra [w_lo] segment (informational: FLOP accounting) */
    const void* weight_ph; /* optional (upsample == 1, 3x3 stride 1 pad 1, Cin % 32 == 0): the four PHASE-SUMMED 2 x 2 kernels of the nearest-2x
                              upsampling conv, slice-major [4 phases (2a + b)][Cin/32][4 taps (2dy + dx)][Cout_pad][32]. Output pixel
                              (2y + a, 2x + b) reads input rows y - 1 + a + dy and columns x - 1 + b + dx with the 3 x 3 taps that land on the same
                              input pixel summed: the conv then runs as four 2 x 2 convolutions of the low-res map (4 / 9 of the MFMA work of
                              gathering nine taps from the virtual high-res map). Same K layout per tap as `weight` (split segments included) | NULL */
    int32_t mx_chunks16;   /* > 0: `in` is an OMGSR_EL_MX operand and `weight_cm` (/ `weight_ph`) its mixed-precision weight: per tap the first
                              mx_chunks16 32-channel chunks are fp16 x fp16, the remaining (Cin / 32 - mx_chunks16) 64-byte chunks hold 64 fp8
                              channels each - first the a_lo' x w_hi' segment, then a_hi' x w_lo' - and run as block-scaled fp8 MFMAs with
                              the E8M0 scales below (weight / operand, per segment). Two shapes: 3x3 stride 1 pad 1 (or the phase form), which
                              always takes the halo-tile kernel; and 1x1 stride 1 (every Linear: `weight` rows are [w_hi fp16 | w_hi' fp8 | w_lo' fp8],
                              K_pad = Cin), which takes igemm_gmx_kernel (ABI v14). Cin = 2 x logical channels (16-bit slots), in_ld = 0. */
    int32_t mx_scale_w1, mx_scale_a1, mx_scale_w2, mx_scale_a2;   /* E8M0 exponents (127 = 2^0): the instruction multiplies each product by 2^(w - 127) 2^(a - 127) */
    int32_t out_mx;        /* 6: ... in the OMGSR_EL_MX6 form (same conditions; only where omgsr_igemm_out_mx6_ok()); 1: the 16-bit output is written in the OMGSR_EL_MX form (4 Cout bytes per pixel; out_dtype OMGSR_OUT_BF16, NHWC,
                              Cout % 64 == 0, fp16 compute type, no GroupNorm statistics): the operand of a following mixed-precision conv */
    int32_t group_tiles;   /* written by omgsr_igemm_multi_plan: halo-kernel tiles of the whole launch group this problem belongs to (0 = alone);
                              kernel choice and the GroupNorm-statistics layout use max(own tiles, group_tiles) */
    uint32_t* overflow_flag; /* optional (fp16 compute type only): a device word the epilogue ORs 1 into when a value it writes as a 16-bit
                              output lies beyond +-65504 (the stores saturate there). The accurate tier's range guard: the pipelines
                              check it once per call, at the sync the reference's forward() already has | NULL.
                              Bit 1 (value 2, diagnostic only): a value beyond +-448 was written into an OMGSR_EL_MX output - its fp8 correction
                              fields saturate (fixed scales), that element keeps single-rounding fp16 accuracy */
    /* ---- ABI v15: GroupNorm apply (+ SiLU) as the conv's patch PRODUCER (SURVEY 2.3 K4 "apply fused into K1 prologue"; replaces the
       pre_norm -> silu -> conv triplets of infer/vaehook.py:269-274, 384-413 / diffusers' ResnetBlock2D norm1 -> nonlinearity -> conv1) */
    const float* gn_scale_shift; /* non-NULL: `in` is the STREAM tensor the GroupNorm reads (element kind `in_el`: the 16-bit compute type or fp32),
                              [N,H,W,Cin] with Cin logical channels, and the kernel normalises while it builds its LDS patch:
                              operand(n, y, x, c) = act(in * scale[n % gn_nimg][c] + shift[n % gn_nimg][c]) rounded once to the compute type,
                              exact zeros outside the image (the conv pads the NORMALISED tensor). Table: f32 [gn_nimg][Cin][2] =
                              (rstd gamma, beta - mean rstd gamma), omgsr_groupnorm_scale_shift(). Only for problems omgsr_igemm_gn_fusable()
                              accepts (3x3 stride 1 pad 1 on the halo-tile kernel's spatial form, plain operand and weight, Cin <= 1024);
                              anything else returns OMGSR_E_SHAPE | NULL */
    int32_t gn_nimg;       /* rows n of `in` share the statistics of image n % gn_nimg (the tiled VAE's tile-major groups); N when every row has its own */
    int32_t gn_act;        /* activation applied after the affine: OMGSR_ACT_SILU (the only one the fused form has: every conv-feeding GroupNorm of the path) */
    int32_t in_el;         /* element kind of `in` when gn_scale_shift is set: OMGSR_EL_16 | OMGSR_EL_F32 */
    int32_t mx_fmt;        /* mx_chunks16 > 0: format of the correction chunks - 0 (or 8): OMGSR_EL_MX (fp8 e4m3, per-tensor scales mx_scale_*);
                              6: OMGSR_EL_MX6 (fp6 e2m3 with one E8M0 scale byte per 32-channel block, in the data; 3x3 convs of the halo-tile
                              kernel only, nine-tap and phase forms). Operand and weight carry the same format. */
} omgsr_igemm_args;
/* 1 when omgsr_igemm / omgsr_igemm_multi would run these arguments with the GroupNorm apply fused into the conv's patch producer (the
 * fields above may still be unset: the answer depends on geometry, operand / weight form, compute type and `in_el` only; for a problem of a
 * launch group call omgsr_igemm_multi_plan first). 0: run omgsr_groupnorm_apply and hand the conv its operand as before. */
int32_t omgsr_igemm_gn_fusable(const omgsr_igemm_args* a);
/* 1 when this problem may write its output as the fp6 operand form (out_mx = 6; the field may still be unset): a 3x3 stride-1 conv that the
 * halo-tile kernel runs in its spatial nine-tap form with a plain / two-term-split fp16 or an OMGSR_EL_MX6 operand, Cout % 64 == 0, no fused
 * statistics (dedicated instantiations: the cooperative fp6 store does not fit next to the other output forms of the shared epilogue). Otherwise
 * the host lets the problem write a stream tensor and runs omgsr_to_operand(y_el = OMGSR_EL_MX6) behind it. */
int32_t omgsr_igemm_out_mx6_ok(const omgsr_igemm_args* a);
/* (scale, shift) table of a GroupNorm for the fused form: out[n][c] = (rstd[n][g] gamma[c], beta[c] - mean[n][g] rstd[n][g] gamma[c]), g = c / (C / G);
 * mean / rstd f32 [nimg][G], gamma / beta f32 [C] | NULL, out f32 [nimg][C][2]. */
int omgsr_groupnorm_scale_shift(const float* mean, const float* rstd, const float* gamma, const float* beta, float* out, int32_t nimg, int32_t C,
                                int32_t G, void* stream);
int omgsr_igemm(const omgsr_igemm_args* a, void* stream);
/* The problems of ONE layer that differ only in tensors and spatial extents (the tiled VAE runs every layer once per tile-shape group:
 * corner / edge / interior tiles are separate dense tensors; infer/vaehook.py:537-829 walks them one tile at a time). Call
 * omgsr_igemm_multi_plan first: it writes `group_tiles` into every problem, so that omgsr_igemm_gn_slots / _gn_entries /
 * _workspace_bytes answer for the group. omgsr_igemm_multi then runs the problems that take the halo-tile kernel in one shape as ONE
 * launch (a prefix table of tile counts in the kernel arguments, <= 8 problems per launch) and everything else in order, one launch
 * each: same results as `count` omgsr_igemm calls, fewer partial last rounds of workgroups. */
int omgsr_igemm_multi_plan(omgsr_igemm_args* args, int32_t count);
int omgsr_igemm_multi(const omgsr_igemm_args* args, int32_t count, void* stream);
/* Batch-invariant dispatch (process-wide, default off). The dispatcher normally picks the kernel family (halo-tile conv vs GEMM-shaped)
 * and a split-K factor from the TOTAL tile count, so the summation order of a layer - and with it the last bits of the result - can change
 * with the batch size. With the flag on those two decisions use `sample_rows`: batch-B == B x batch-1 bit for bit (the reference's
 * contract), at the cost of under-filled launches for small batches. */
int omgsr_set_batch_invariant(int on);
/* Bytes of `workspace` that would let omgsr_igemm split the contraction of a small-M / large-K problem over
 * several workgroups (fp32 partial tiles + a reduce pass that applies the epilogue); 0 = no split for this shape. */
int64_t omgsr_igemm_workspace_bytes(const omgsr_igemm_args* a);
/* Slots per image / entries per slot of `gn_partial` if omgsr_igemm can emit the GroupNorm statistics of its output for
 * these arguments (gn_groups set; NHWC 16-byte rows, no GEGLU, batch 1; halo-tile path: one slot per wave tile; GEMM-shaped
 * kernels: one slot per 32-row block when Ho*Wo % 32 == 0; not on the split-K path), else 0: the caller then runs
 * omgsr_groupnorm_stats on the output instead. */
int32_t omgsr_igemm_gn_slots(const omgsr_igemm_args* a);
int32_t omgsr_igemm_gn_entries(const omgsr_igemm_args* a);

/*
 * K4 — GroupNorm statistics and apply (replaces F.group_norm; the externally supplied
 * (mean, var) form is what infer/vaehook.py:384-413 custom_group_norm needs).
 * stats: x bf16 [N, HW, C]; partial f32 [N][nchunk][G][2] workspace; mean/rstd f32 [N][G].
 * nchunk = omgsr_groupnorm_nchunk(HW).  `var_out` (optional) receives the biased variance.
 */
int omgsr_groupnorm_nchunk(int64_t HW);
int omgsr_groupnorm_stats(const void* x, float* partial, float* mean, float* rstd, float* var_out,
                          int32_t N, int64_t HW, int32_t C, int32_t G, float eps, int32_t x_el, void* stream);
/* Second half of omgsr_groupnorm_stats alone: fold partial [N][nslot][G][2] (from omgsr_igemm's gn_partial) into
 * mean / rstd (/ biased variance); count = elements per (image, group) = HW * C / G. */
int omgsr_groupnorm_finalize(const float* partial, float* mean, float* rstd, float* var_out, int32_t N,
                             int32_t nslot, int32_t G, int32_t entries, double count, float eps, void* stream);
/* First half of omgsr_groupnorm_stats alone: partial [N][omgsr_groupnorm_nchunk(HW)][G][2]. */
int omgsr_groupnorm_partial(const void* x, float* partial, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t x_el, void* stream);
/*
 * Tiled-VAE statistics (infer/vaehook.py:459-534 GroupNormParam.summary + :384-413 custom_group_norm): the tiles of
 * one image come in up to OMGSR_GN_MAX_GROUPS shape groups; group k holds tiles[k] x N rows (tile-major) of partials
 * [rows][nslot[k]][G][2] with count[k] elements per (row, group channel set). Per image:
 *   mean[n,g] = sum_k weight[k] * sum_t mean_{k,t,n,g},  var likewise,  rstd = rsqrt(var + eps)
 * (weight[k] = pixels of one tile of group k / pixels of all tiles: the reference's pixel-weighted merge).
 */
#define OMGSR_GN_MAX_GROUPS 8
typedef struct omgsr_gn_merge_args {
    const float* partial[OMGSR_GN_MAX_GROUPS];
    double count[OMGSR_GN_MAX_GROUPS];
    float weight[OMGSR_GN_MAX_GROUPS];
    int32_t tiles[OMGSR_GN_MAX_GROUPS];
    int32_t nslot[OMGSR_GN_MAX_GROUPS];
    int32_t entries[OMGSR_GN_MAX_GROUPS];   /* entries per slot of partial[k]: G or C */
    int32_t ngroups;
} omgsr_gn_merge_args;
int omgsr_groupnorm_finalize_merged(const omgsr_gn_merge_args* a, float* mean, float* rstd, float* var_out,
                                    int32_t N, int32_t G, float eps, void* stream);
/* y = act((x - mean[n,g]) * rstd[n,g] * gamma[c] + beta[c]);  act in {NONE, SILU}. x and y may alias when both are
 * OMGSR_EL_16. x_el: OMGSR_EL_16 | OMGSR_EL_F32; y_el: OMGSR_EL_16 | OMGSR_EL_SPLIT (y is an MFMA operand).
 * y2 (optional, x_el OMGSR_EL_F32 only): a second output, x ITSELF rounded to an operand of kind y2_el (OMGSR_EL_16 |
 * OMGSR_EL_SPLIT) - the input of the ResnetBlock's 1x1 conv_shortcut, cast while the tensor streams by.
 * overflow_flag (optional, fp16 compute type): ORed with 1 when a y2 value lies beyond +-65504 (omgsr_igemm_args.overflow_flag). */
int omgsr_groupnorm_apply(const void* x, void* y, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int32_t N, int64_t HW, int32_t C,
                          int32_t G, int32_t act, int32_t x_el, int32_t y_el, void* y2, int32_t y2_el, uint32_t* overflow_flag, void* stream);
/* Same with `rows` rows of x sharing `stat_rows` rows of statistics: row r uses mean[r % stat_rows] (tile-major tiles). */
int omgsr_groupnorm_apply_shared(const void* x, void* y, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, int32_t rows, int64_t HW, int32_t C,
                                 int32_t G, int32_t act, int32_t stat_rows, int32_t x_el, int32_t y_el, void* y2, int32_t y2_el,
                                 uint32_t* overflow_flag, void* stream);

/*
 * K9/K10 — LayerNorm over the last dim (replaces F.layer_norm and the AdaLN-Zero modulate chain of
 * FluxTransformerBlock): y = (x - mu) * rsqrt(var + eps) * a[c] + b[c], a/b f32 [C] or NULL (1 / 0).
 */
int omgsr_layernorm(const void* x, void* y, const float* a, const float* b, int64_t rows, int32_t C,
                    float eps, int32_t x_el, int32_t y_el, void* stream);
/* Stream tensor -> MFMA operand: y = x rounded to the compute type (y_el OMGSR_EL_16) or its two-term split
 * (OMGSR_EL_SPLIT); x f32 [rows][C], C % 8 == 0. (Inputs of convs that no norm precedes: up / down-sampling convs,
 * 1x1 shortcuts, conv_in, proj_out, post_quant_conv.) */
int omgsr_to_operand(const float* x, void* y, int64_t rows, int32_t C, int32_t y_el, uint32_t* overflow_flag, void* stream);
/* ABI v17: V of an attention as the TRANSPOSED two-term split (omgsr_attn_args.vt_lo_off): x f32 [B][L][C] (a projection's stream output) ->
 * y 16-bit [B][2C][ld], y[b][c][l] = round(x[b][l][c]), y[b][C + c][l] = round(x - hi); ld >= L, ld % 8 == 0, columns >= L are left untouched
 * (the caller zero-fills). Replaces v.transpose(1, 2) in fp32 under F.scaled_dot_product_attention for the range-fallback tier. */
int omgsr_transpose_split(const float* x, void* y, int32_t B, int32_t L, int32_t C, int64_t ld, void* stream);

/*
 * K7/K8 — fused softmax(Q K^T * scale) V on MFMA (replaces F.scaled_dot_product_attention).
 * q  bf16 rows [B][Lq]  with row stride q_ld,  head h at column h*D
 * k  bf16 rows [B][Lk]  with row stride k_ld
 * vt bf16 V TRANSPOSED: [B][H*D][vt_ld], key index contiguous (written by omgsr_igemm LAYOUT_T)
 * o  bf16 rows [B][Lq]  with row stride o_ld
 * D in {64, 128}; Lk arbitrary (keys >= Lk are masked); kv_bstride 0 broadcasts one K/V to all B.
 */
typedef struct omgsr_attn_args {
    const void* q; const void* k; const void* vt; void* o;
    int32_t B, H, D, Lq, Lk;
    int64_t q_ld, k_ld, vt_ld, o_ld;
    int64_t q_bstride, k_bstride, vt_bstride, o_bstride;
    float scale;
    int32_t o_lo_off;      /* > 0: o is written as the two-term split: lo lands o_lo_off columns after hi
                              (o_lo_off >= H*D, o_ld >= o_lo_off + H*D, o_lo_off % 4 == 0) */
    int32_t o_mx;          /* 1: o is written in the mixed-precision operand form OMGSR_EL_MX (the output projection is an MX GEMM,
                              omgsr_igemm with mx_chunks16 > 0): a row of C = H*D channels is 4C bytes [hi fp16 | lo' fp8 | hi' fp8];
                              o_ld = 2C (16-bit slots), o_lo_off = 0, C % 64 == 0, fp16 compute type (ABI v14) */
    int32_t q_lo_off;      /* ABI v17, with k_lo_off (both > 0 or both 0; D = 64): q and k are TWO-TERM SPLITS - the low halves of a row sit q_lo_off /
                              k_lo_off elements after q / k (a fused [q | k] projection written with out_lo_off: [q_hi | k_hi | q_lo | k_lo]); the
                              scores are K_hi Q_hi^T + K_lo Q_hi^T + K_hi Q_lo^T in one fp32 accumulator. The range-fallback tier's bf16 operands
                              (8-bit mantissas) need it to hold the north-star tolerance on the UNet (replaces nothing in torch: F.sdpa in fp32) */
    int32_t k_lo_off;
    int32_t p_split;       /* 1 (with the q / k split): the probabilities enter O^T = V^T P^T as p_hi + p_lo (both 16-bit, from the fp32 value in
                              registers): one more pass of the PV MFMAs, no memory traffic */
    int32_t reserved1;
    int64_t vt_lo_off;     /* > 0 (with the q / k split): V^T is a two-term split too - the low halves of [H*D][vt_ld] start vt_lo_off ELEMENTS after
                              vt (omgsr_transpose_split writes [B][2 H D][ld]: vt_lo_off = H * D * vt_ld, vt_bstride = 2 H D vt_ld); O^T gains the
                              V_lo^T P_hi^T pass */
} omgsr_attn_args;
/* Process-wide (default 0): the online softmax moves its running maximum only when a row's maximum grows by more than 2^t in
 * the scaled base-2 domain (probabilities then reach 2^t instead of 1; the result is the same quotient). 0 = exact running
 * maximum. 0 <= t <= 12. +5 % attention throughput at t = 8; the fast tiers use it, the accurate tier does not. */
int omgsr_set_attention_defer_max(float log2_threshold);
int omgsr_attention(const omgsr_attn_args* a, void* stream);

/* Row softmax for the unfused d=512 VAE attention: p = softmax(s[:, :Lvalid]); s f32 [rows][L],
 * p bf16 [rows][L]; columns >= Lvalid (zero-padded keys) come out as exactly 0. */
int omgsr_softmax_rows(const float* s, void* p, int64_t rows, int32_t L, int32_t Lvalid, void* stream);
/* ABI v17: the same with the probabilities as a two-term split, p [rows][2 L] = [p_hi | p_lo] (range-fallback tier: the first factor of the
 * PV product, omgsr_igemm with in_ld = L and a [v_hi | v_hi | v_lo] second factor). */
int omgsr_softmax_rows_split(const float* s, void* p, int64_t rows, int32_t L, int32_t Lvalid, void* stream);

/*
 * K11 — RMSNorm(q,k over head_dim) * w then interleaved-pair RoPE, in place
 * (FluxAttnProcessor2_0: norm_q/norm_k + apply_rotary_emb).  x bf16 rows [B*L] with stride ld,
 * head h at column col0 + h*D; w f32 [H][D] (per-head weight rows, so one call covers the q heads and the
 * k heads of a fused [q|k] buffer); cos/sin f32 [>= pos0+L][D] (repeat-interleaved pairs); rope may be NULL.
 * w2 (optional): a second weight table used by the rows whose position in their sequence (row % L) is >= Lsplit - the joint
 * [text ; image] sequence of a FluxTransformerBlock (norm_added_q/k for the text rows, norm_q/k for the image rows) in one call.
 */
int omgsr_rmsnorm_rope(void* x, const float* w, const float* w2, int32_t Lsplit, const float* cos_t, const float* sin_t, int32_t B,
                       int32_t L, int32_t H, int32_t D, int64_t ld, int32_t col0, int32_t pos0,
                       float eps, void* stream);

/* K14 — layout and latent algebra. */
/* NCHW (f32 or bf16 per src_dtype: 0 bf16, 1 f32) -> NHWC bf16 with channels zero-padded to Cpad. */
int omgsr_nchw_to_nhwc(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                       int32_t Cpad, int32_t src_dtype, int32_t dst_el, void* stream);
/* NHWC bf16 (row stride ld, first C channels) -> NCHW (dst_dtype 0 bf16, 1 f32); optional clamp to [lo,hi]. */
int omgsr_nhwc_to_nchw(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                       int32_t ld, int32_t dst_dtype, int32_t do_clamp, float lo, float hi, int32_t src_el, void* stream);
/* dst[..., off:off+C] = src (bf16 rows) — channel concat building block (torch.cat(dim=1) in NCHW). */
int omgsr_copy_channels(const void* src, void* dst, int64_t rows, int32_t C, int32_t src_ld,
                        int32_t dst_ld, int32_t dst_off, int32_t el, void* stream);
/*
 * DiagonalGaussianDistribution.sample() * scale (infer/omgsr_s_infer_model.py:173,
 * infer/omgsr_f_infer_model.py:16-17): z = ((mu + exp(0.5*clamp(logvar,-30,20)) * eps) - shift) * scale.
 * moments bf16 NHWC [rows][2*C]; eps f32 NHWC [rows][C]; z bf16 NHWC [rows][ld_out] (cols >= C zeroed).
 */
int omgsr_vae_sample(const void* moments, const float* eps, void* z, int64_t rows, int32_t C,
                     int32_t ld_out, float shift, float scale, int32_t el, void* stream);
/* out = (x * a + y * b + c) * d, bf16 tensors with bf16 rounding after every op when `bf16_steps`
 * (mirrors the reference's eager bf16 arithmetic, infer/omgsr_s_infer_model.py:80-84). */
int omgsr_axpby(const void* x, const void* y, void* out, int64_t n, float a, float b, float c, float d,
                int32_t bf16_steps, int32_t el, void* stream);
/* ABI v16: y f32[rows][N] = act(x f32[rows][K]) . w f32[N][K]^T + b f32[N] (b may be NULL; act = SiLU when `silu_in`), fp32 FMAs in a fixed
 * order (one wave per output element). The load-time constant folds: diffusers TimestepEmbedding / CombinedTimestep(Guidance)TextProjEmbeddings,
 * AdaLayerNormZero(.Single) / AdaLayerNormContinuous `linear(silu(temb))`, ResnetBlock2D `time_emb_proj(silu(temb))` - all functions of
 * (t*, guidance, prompt) only (infer/omgsr_s_infer_model.py:107-110, infer/omgsr_f_infer_model.py:83-88). */
int omgsr_linear_f32(const float* x, const float* w, const float* b, float* y, int32_t rows, int32_t K, int32_t N,
                     int32_t silu_in, void* stream);
/* acc[n,y0+y,x0+x,c] += tile[n,y,x,c] * w[y,x] (f32 acc, bf16 tile NHWC ld=tile_ld), and the matching
 * normaliser; infer/omgsr_s_infer_model.py:137-161. */
int omgsr_tile_accumulate(const void* tile, const float* w, float* acc, int32_t N, int32_t C,
                          int32_t th, int32_t tw, int32_t tile_ld, int32_t H, int32_t W, int32_t y0,
                          int32_t x0, int32_t tile_el, void* stream);
/* out bf16[rows][ld] = acc f32[rows][C] / wsum[pixel]  (cols >= C zeroed) */
int omgsr_tile_normalise(const float* acc, const float* wsum, void* out, int32_t N, int64_t HW,
                         int32_t C, int32_t ld, int32_t out_el, void* stream);
/* bf16 window copy: dst[n,y,x,:] = src[n,y0+y,x0+x,:] */
int omgsr_crop_nhwc(const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t y0,
                    int32_t x0, int32_t th, int32_t tw, int32_t el, void* stream);
/* bf16 window paste: dst[n, dy0+y, dx0+x, :] = src[n, sy0+y, sx0+x, :] for y < th, x < tw (tiled-VAE
 * crop_valid_region + result[...] = tile, infer/vaehook.py:416-427,805). */
int omgsr_paste_nhwc(const void* src, void* dst, int32_t N, int32_t C, int32_t sH, int32_t sW, int32_t sy0,
                     int32_t sx0, int32_t dH, int32_t dW, int32_t dy0, int32_t dx0, int32_t th, int32_t tw,
                     int32_t el, void* stream);
/* ABI v17: F.interpolate(x, scale_factor = s, mode = "nearest-exact") on an NHWC tensor (the fast tiled-VAE mode's down-sampled copy,
 * infer/vaehook.py:714-735): dst[n][y][x] = src[n][min(floorf((y + .5) scale_y), H - 1)][min(floorf((x + .5) scale_x), W - 1)] with
 * scale_* = (float)(1 / s) (ATen's nearest_exact_idx); Ho = floor(H s), Wo = floor(W s) are the caller's. C % 8 == 0 (% 4 for OMGSR_EL_F32). */
int omgsr_resize_nearest_exact_nhwc(const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                                    float scale_y, float scale_x, int32_t el, void* stream);
/* Flux 2x2 pack / unpack between NHWC [N,H,W,C(ld)] and tokens [N,(H/2)(W/2),4C] with channel
 * order c*4 + dy*2 + dx (infer/omgsr_f_infer_model.py:21-41). dir 0 = pack, 1 = unpack. */
int omgsr_flux_pack(const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ld,
                    int32_t dir, int32_t el, void* stream);

/*
 * SURVEY §8(f) f1 — the driver's post-process on the device (replaces infer/infer_omgsr_s.py:96-103 and
 * infer/wavelet_color_fix.py:12-125, which run on the CPU through PIL per image):
 *   target = ToPILImage(clip(sr * 0.5 + 0.5, 0, 1))   (the "+ 0.5" rounds to the model dtype, the byte conversion truncates)
 *   out    = uint8( colour_fix(target / 255, source / 255).clamp(0, 1) * 255 )
 * sr_nhwc: model output, 16-bit compute dtype, [N,H,W,sr_ld] (channels 0..2 used), values in [-1, 1] unclamped;
 * src_hwc3 / out_hwc3: uint8 [N,H,W,3] (the upscaled LQ image the model was fed / the result, PIL memory order).
 * method: NONE (plain conversion, src and workspace may be NULL), ADAIN (per-image per-channel mean / unbiased-std
 * match, statistics integer-exact), WAVELET (5-level a-trous low-frequency swap). workspace: device scratch of
 * omgsr_colorfix_workspace_bytes() bytes.
 */
/* uint8 [N,H,W,3] -> model input [N,H,W,8] (compute dtype, channels 3..7 zero): to_tensor(img).to(dtype) * 2 - 1
 * (infer/infer_omgsr_s.py:92), rounded to the dtype after each torch op. */
int omgsr_image_to_model_input(const uint8_t* img_hwc3, void* out_nhwc8, int32_t N, int32_t H, int32_t W, int32_t out_el, void* stream);
#define OMGSR_COLORFIX_NONE 0
#define OMGSR_COLORFIX_ADAIN 1
#define OMGSR_COLORFIX_WAVELET 2
int64_t omgsr_colorfix_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t method);
int omgsr_colorfix(const void* sr_nhwc, int32_t sr_ld, const uint8_t* src_hwc3, uint8_t* out_hwc3, void* workspace,
                   int32_t N, int32_t H, int32_t W, int32_t method, int32_t sr_el, void* stream);

/*
 * SURVEY §8(f) f2 — the driver's PRE-process on the device (replaces the PIL resizes of infer/infer_omgsr_s.py:71-84:
 * `Image.resize` BICUBIC x upscale, then LANCZOS to width / height multiples of 8): ONE pass of Pillow's 8-bit resampler
 * (libImaging/Resample.c ImagingResampleHorizontal / Vertical_8bpc), bit for bit.
 *   src u8 [N,Hin,Win,3] (PIL memory order)  ->  dst u8 [N,Hout,Wout,3]; axis 1: horizontal (Wout = out_size, Hout = Hin),
 *   axis 0: vertical (Hout = out_size, Wout = Win). Per output index o along the axis: bounds[2o] = first input index,
 *   bounds[2o+1] = tap count (<= ksize), kk[o*ksize + t] = weight of tap t in 2^22 fixed point (device int32 tables, computed
 *   on the host in float64 like Pillow: omgsr_amd/preprocess.py). dst = clip8((2^21 + sum(src * kk)) >> 22).
 * Image.resize = the horizontal pass, then the vertical pass on its 8-bit result; a pass whose size does not change is skipped.
 */
int omgsr_resample_u8(const uint8_t* src, uint8_t* dst, const int32_t* bounds, const int32_t* kk, int32_t ksize,
                      int32_t N, int32_t Hin, int32_t Win, int32_t out_size, int32_t axis, void* stream);

/* Optional per-launch timing (HIP events on the launch stream) for bench.py's roofline leg. */
int omgsr_timing_enable(int on);
int omgsr_timing_reset(void);
/* Synchronises, then fills up to `cap` entries; returns the number of recorded launches. */
/* kind: 1 igemm, 2 attention, 3 groupnorm, 4 layernorm, 5 elementwise, 6 softmax. variant (igemm only): which kernel
 * the dispatcher launched - 1 igemm_kernel (register staged), 2 igemm_dma_kernel, 3 igemm_halo_kernel, 4 igemm_dma_kernel
 * split-K + splitk_reduce_kernel, 5 igemm_p8_kernel, 6 igemm_halo_kernel in its phase-decomposed upsampling form, 7 / 8
 * igemm_halo_multi_kernel (several problems of one layer in one launch: nine-tap / phase-decomposed form). flops / bytes: ALGORITHMIC work of the launch (a two-term split operand's duplicated
 * channels count once). */
typedef struct omgsr_timing_entry { int32_t kind; float ms; double flops; double bytes; int64_t m, n, k; int32_t variant; int32_t stage; } omgsr_timing_entry;
int omgsr_timing_collect(omgsr_timing_entry* out, int cap);
/* Tag every launch recorded from now on with `stage` (the pipelines mark encode = 1 / denoiser = 2 / decode = 3; 0 = untagged), so the
 * roofline leg can report per-stage time and the denoiser-only MFMA fraction. Free when timing is off. */
int omgsr_timing_stage(int stage);
/* Dense 16-bit MFMA micro-benchmark (compute dtype) on the current device: measured TFLOP/s through the HOST pointer.
 * Synchronises `stream`; a benchmarking utility (bench.py records it beside the roofline fractions). */
int omgsr_mfma_peak(int32_t iters, float* tflops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMGSR_HIP_H */
