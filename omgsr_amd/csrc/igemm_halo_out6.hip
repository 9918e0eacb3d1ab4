// Halo-tile kernel instantiations whose epilogue writes the fp6 operand form (OMGSR_EL_MX6, omgsr_igemm_args.out_mx = 6) and nothing else: the
// ResnetBlock conv2 in front of an up-sampler, whose only consumer is the up-sampler's phase-form conv (round 5). Two operand forms - plain /
// two-term-split fp16 (the decoder's single layers) and fp6 corrections (MX = 6) - as single launches and launch groups; spatial nine-tap form,
// fp16 compute type, no fused statistics (out_mx excludes gn_partial). A translation unit of its own: as a run-time option of the shared
// epilogue the cooperative store cost every kernel 600+ bytes of scratch (profiles/r05_experiments.md).
#include "igemm_halo_body.hip.h"

namespace omgsr {
static int out6_attrs() {
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, 0, 0, 0, true>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, 6, 0, 0, true>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, 0, 0, 0, false, true>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, 6, 0, 0, false, true>)};
        const int rc = halo_set_lds_attr(fns, 4);
        if (rc != 0) return rc;
        attr_set = true;
    }
    return 0;
}
// what these instantiations can run (omgsr_igemm_out_mx6_ok: the host asks before it requests out_mx = 6)
bool igemm_halo_out6_ok(const omgsr_igemm_args& a) {
    return compute_dtype() == 1 && a.R == 3 && a.S == 3 && a.stride == 1 && !a.upsample && !a.gn_scale_shift && !a.gn_partial && a.Cout > 32 && (a.Cout & 63) == 0 &&
           (a.mx_chunks16 == 0 || a.mx_fmt == 6);          // (never FLAT: halo_flat_eligible() refuses out_mx = 6 problems)
}
int igemm_halo_out6_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    const int rc = out6_attrs();
    if (rc != 0) return rc;
    halo_geo(a, g, false);
    if (g.flat) return OMGSR_E_SHAPE;
    if (a.mx_chunks16 > 0) hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, 6, 0, 0, true>), dim3(g.ntm * g.ntn), dim3(256), LDS_BYTES, st, a, g);
    else hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, 0, 0, 0, true>), dim3(g.ntm * g.ntn), dim3(256), LDS_BYTES, st, a, g);
    return (int)hipGetLastError();
}
int igemm_halo_out6_launch_multi(const void* halo_multi, unsigned blocks, hipStream_t st) {
    const HaloMulti& m = *reinterpret_cast<const HaloMulti*>(halo_multi);
    const int rc = out6_attrs();
    if (rc != 0) return rc;
    if (m.p[0].mx_chunks16 > 0) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, 6, 0, 0, false, true>), dim3(blocks), dim3(256), LDS_BYTES, st, m);
    else hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, 0, 0, 0, false, true>), dim3(blocks), dim3(256), LDS_BYTES, st, m);
    return (int)hipGetLastError();
}
}  // namespace omgsr
