// fp6 instantiations of the halo-tile kernel body (igemm_halo_body.hip.h, MX = 6; round 5): fp16 chunks followed by correction chunks of
// fp6 (e2m3) codes with per-32-channel E8M0 scales taken from the data (OMGSR_EL_MX6 operand + weight packed the same way): the f8f6f4 MFMA
// in its 8-pass form, 1.5x the plain fp16 matrix-pipe time per layer where the fp8 form (igemm_halo_mx.hip) takes 2x. Spatial forms (nine taps, and the four-tap phase form of the up-sampling convs, whose operand igemm_halo_out6.hip / the cast kernel writes).
#include "igemm_halo_body.hip.h"

namespace omgsr {
static int mx6_attrs() {
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, 6>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, 6>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 4, 6>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 4, 6>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, 6, 0, 0, true>)};      // split-K launch groups
        const int rc = halo_set_lds_attr(fns, 5);
        if (rc != 0) return rc;
        attr_set = true;
    }
    return 0;
}
int igemm_halo_launch_multi_mx6(const void* halo_multi, unsigned blocks, hipStream_t st) {
    const HaloMulti& m = *reinterpret_cast<const HaloMulti*>(halo_multi);
    const int rc = mx6_attrs();
    if (rc != 0) return rc;
    if (m.p[0].upsample) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 4, 6>), m.g[0].interleave ? dim3(4 * blocks) : dim3(blocks, 4), dim3(256), LDS_BYTES, st, m);
    else if (m.g[0].cc1 > 0) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, 6, 0, 0, true>), dim3(blocks), dim3(256), LDS_BYTES, st, m);
    else hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, 6>), dim3(blocks), dim3(256), LDS_BYTES, st, m);
    return (int)hipGetLastError();
}
// g: after halo_geo, spatial form (g.flat == 0)
int igemm_halo_launch_mx6(const omgsr_igemm_args& a, const IgemmGeo& g, hipStream_t st) {
    const int rc = mx6_attrs();
    if (rc != 0) return rc;
    if (a.upsample) hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 4, 6>), g.interleave ? dim3(32 * ((g.ntm * g.ntn + 7) / 8)) : dim3(g.ntm * g.ntn, 4), dim3(256), LDS_BYTES, st, a, g);
    else hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, 6>), dim3(g.ntm * g.ntn), dim3(256), LDS_BYTES, st, a, g);
    return (int)hipGetLastError();
}
}  // namespace omgsr
