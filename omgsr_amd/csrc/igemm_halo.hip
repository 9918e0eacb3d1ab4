// 3x3 stride-1 pad-1 convolution, halo-tile variant of the LDS-DMA implicit GEMM (gfx950).
//
// profiles/r01_pmc_igemm.md: the LDS-DMA GEMM kernel spends ~70 % of its time just streaming operands
// L2 -> LDS (24 KB per 2.1 MFLOP K-step); 9 of every 10 A bytes are the SAME input pixels fetched again
// for another tap. This kernel cuts the bytes instead of chasing the stream:
//
//  * an output tile is SPATIAL: 8 rows x 32 pixels (256 outputs) x 128 channels; for one 32-channel chunk
//    the (8+2) x (32+2) input patch is DMA'd into LDS ONCE (21.3 KB) and all 9 taps read their MFMA A
//    fragments from it at shifted rows (fragment = 32 consecutive pixels of one tile row, so the swizzled
//    linear image stays conflict-free at any shift)      -> A traffic 144 KB -> 24 KB per chunk
//  * K order is chunk-major: k = (cc*9 + tap)*32 + c, and the weights are packed SLICE-major (`weight_cm` =
//    [Cin/32][9][Cout_pad][32]): a K-step = (chunk, tap) streams its 8 KB weight slice, one contiguous run of
//    full 128-B lines, through a 3-deep ring (row-major weights made every DMA instruction touch 16 half lines)
//  * per 18.9 MFLOP chunk: 24 KB (A, double-buffered, prefetched one chunk ahead) + 72 KB (B) = 5.1 KB/MFLOP
//    vs 12 KB/MFLOP for the GEMM-shaped kernel
//  * 4 waves, 128 x 64 per wave (4 tile rows x 64 couts), one raw s_barrier per K-step, counted vmcnt
//  * image borders / ragged W,H: patch pixels outside the image come from the zero page, output columns
//    past W and rows past H are dropped in the epilogue (tiled-VAE tiles are 86, 172, 320 ... wide)
//  * TAPS = 4: nearest-2x upsampling + 3x3 conv WITHOUT the redundant taps. Output pixel (2y + a, 2x + b) of conv3x3(up2(in)) only
//    ever sees the 2 x 2 input pixels rows {y - 1 + a, y + a} x cols {x - 1 + b, x + b}: taps that land on the same input pixel
//    have their weights SUMMED at pack time (phase a = 0: row y - 1 <- w[0], row y <- w[1] + w[2]; a = 1: row y <- w[0] + w[1],
//    row y + 1 <- w[2]; same along x), so each of the four output phases is a 2 x 2 convolution of the LOW-resolution map: 4 / 9 of
//    the MFMA work of the gather form. One launch, blockIdx.y = phase; the tile is 8 x 32 low-res pixels whose outputs land at
//    stride 2 in the high-res map; patch (8+1) x (32+1), weight ring of 4 stages (stage = tap).
#include "igemm_halo_body.hip.h"

// This translation unit launches ONE problem per grid (igemm_halo_kernel). The several-problems-per-launch and mixed-precision
// instantiations of the same body live in igemm_halo_multi.hip (two files so that hipcc compiles them in parallel).
namespace omgsr {
// Preconditions (checked by the dispatcher): R = S = 3, stride 1, pad 1 (on the virtual, optionally 2x-upsampled input), Cin % 32 == 0,
// weight_cm != NULL, batch == 1, W >= 16.
static int prio_min_cin() {
    static const char* e = getenv("OMGSR_HALO_PRIO_CIN");      // A/B runs
    static const int v = e ? atoi(e) : (1 << 30);      // default: never. +6..10 % on Cin >= 512 before the slice-major weights, -0.5 % of the S-1024 step after
    return v;
}

int igemm_halo_launch_mx(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st);
int igemm_halo_flat_launch(const omgsr_igemm_args& a, const IgemmGeo& g, hipStream_t st);       // igemm_halo_flat.hip
int igemm_halo_launch_f16(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st, bool phase, bool narrow);      // igemm_halo_f16.hip

// one problem per grid, bf16 compute type (the fp16 instantiations are a translation unit of their own: hipcc compiles the files in parallel)
int igemm_halo_gn_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st);        // igemm_halo_gn.hip

int igemm_halo_out6_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st);              // igemm_halo_out6.hip

int igemm_halo_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st, const bool phase) {
    if (a.out_mx == 6) return phase ? OMGSR_E_SHAPE : igemm_halo_out6_launch(a, g, st);
    if (a.gn_scale_shift) return phase ? OMGSR_E_SHAPE : igemm_halo_gn_launch(a, g, st);
    if (a.mx_chunks16 > 0) return igemm_halo_launch_mx(a, g, st);
    const bool narrow = halo_geo(a, g, phase);
    if (g.flat) return igemm_halo_flat_launch(a, g, st);
    if (omgsr::compute_dtype() == 1) return igemm_halo_launch_f16(a, g, st, phase, narrow);
    using T = bf16_t;
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(igemm_halo_kernel<T, 0, false>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<T, 0, true>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<T, 1, false>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<T, 2, false>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<T, 3, false>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<T, 0, false, true>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<T, 5, false>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<T, 0, false, false, 4>)};
        const int rc = halo_set_lds_attr(fns, (int)(sizeof(fns) / sizeof(fns[0])));
        if (rc != 0) return rc;
        attr_set = true;
    }
    const dim3 grid = (phase && g.interleave) ? dim3(32 * ((g.ntm * g.ntn + 7) / 8)) : dim3(g.ntm * g.ntn, phase ? 4 : 1, 1);
    static const char* abl = ablation_env("OMGSR_HALO_ABLATE");      // timing experiments only: results are garbage (needs OMGSR_ABLATION_OK=1; bf16 only)
    static const char* var = getenv("OMGSR_HALO_VARIANT");     // A/B runs: "0" = LDS-DMA issued in front of the step's MFMAs
    if (phase) hipLaunchKernelGGL((igemm_halo_kernel<T, 0, false, false, 4>), grid, dim3(256), LDS_BYTES, st, a, g);
    else if (var && var[0] == '0' && !narrow) hipLaunchKernelGGL((igemm_halo_kernel<T, 5, false>), grid, dim3(256), LDS_BYTES, st, a, g);
    else if (abl && abl[0] == '1') hipLaunchKernelGGL((igemm_halo_kernel<T, 1, false>), grid, dim3(256), LDS_BYTES, st, a, g);
    else if (abl && abl[0] == '2') hipLaunchKernelGGL((igemm_halo_kernel<T, 2, false>), grid, dim3(256), LDS_BYTES, st, a, g);
    else if (abl && abl[0] == '3') hipLaunchKernelGGL((igemm_halo_kernel<T, 3, false>), grid, dim3(256), LDS_BYTES, st, a, g);
    else if (narrow) hipLaunchKernelGGL((igemm_halo_kernel<T, 0, false, true>), grid, dim3(256), LDS_BYTES, st, a, g);
    else if (a.Cin >= prio_min_cin()) hipLaunchKernelGGL((igemm_halo_kernel<T, 0, true>), grid, dim3(256), LDS_BYTES, st, a, g);
    else hipLaunchKernelGGL((igemm_halo_kernel<T, 0, false>), grid, dim3(256), LDS_BYTES, st, a, g);
    return (int)hipGetLastError();
}

int igemm_halo_gn_slots(const omgsr_igemm_args& a, const bool phase) {
    if (phase) return 2 * 4 * ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    if (const int P = halo_flat_pitch(a)) return 2 * ((a.Ho * P + TH * TW - 1) / (TH * TW));
    return 2 * ((a.Wo + TW - 1) / TW) * ((a.Ho + TH - 1) / TH);
}
int igemm_halo_tiles(const omgsr_igemm_args& a, const bool phase) {
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    if (phase) return a.N * ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH) * ((logical_cols + BN - 1) / BN) * 4;
    if (const int P = halo_flat_pitch(a)) return a.N * ((a.Ho * P + TH * TW - 1) / (TH * TW)) * ((logical_cols + BN - 1) / BN);
    return a.N * ((a.Wo + TW - 1) / TW) * ((a.Ho + TH - 1) / TH) * ((logical_cols + BN - 1) / BN);
}
int igemm_halo_flat(const omgsr_igemm_args& a) { return halo_flat_pitch(a); }
// tiles of the problem in ONE given form (0: spatial, 1: FLAT; -1 when the FLAT form cannot run it): what omgsr_igemm_multi_plan sums over a group
int igemm_halo_tiles_form(const omgsr_igemm_args& a, const int flat) {
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    if (flat && !halo_flat_eligible(a)) return -1;
    return a.N * (flat ? halo_flat_tiles_per_image(a) : halo_grid_tiles_per_image(a)) * ((logical_cols + BN - 1) / BN);
}
}  // namespace omgsr
