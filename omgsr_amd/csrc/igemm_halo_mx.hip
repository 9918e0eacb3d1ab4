// Mixed-precision instantiations of the halo-tile kernel body (igemm_halo_body.hip.h, MX = true): fp16 chunks followed by block-scaled
// fp8 chunks (v_mfma_scale_f32_32x32x64_f8f6f4) of an OMGSR_EL_MX operand; fp16 compute type, wide nine-tap shape.
#include "igemm_halo_body.hip.h"

namespace omgsr {
static int mx_attrs() {
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, true>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, true>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 4, true>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 4, true>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, true, 0, 0, true>)};      // split-K launch groups
        const int rc = halo_set_lds_attr(fns, 5);
        if (rc != 0) return rc;
        attr_set = true;
    }
    return 0;
}
int igemm_halo_launch_multi_mx6(const void* halo_multi, unsigned blocks, hipStream_t st);       // igemm_halo_mx6.hip
int igemm_halo_launch_mx6(const omgsr_igemm_args& a, const IgemmGeo& g, hipStream_t st);
int igemm_halo_launch_multi_mx(const void* halo_multi, unsigned blocks, hipStream_t st) {
    const HaloMulti& m = *reinterpret_cast<const HaloMulti*>(halo_multi);         // igemm_halo_multi.hip's struct of the same header
    if (m.p[0].mx_fmt == 6) return igemm_halo_launch_multi_mx6(halo_multi, blocks, st);
    const int rc = mx_attrs();
    if (rc != 0) return rc;
    if (m.p[0].upsample) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 4, true>), m.g[0].interleave ? dim3(4 * blocks) : dim3(blocks, 4), dim3(256), LDS_BYTES, st, m);
    else if (m.g[0].cc1 > 0) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, true, 0, 0, true>), dim3(blocks), dim3(256), LDS_BYTES, st, m);      // chunk ranges of one problem
    else hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, true>), dim3(blocks), dim3(256), LDS_BYTES, st, m);
    return (int)hipGetLastError();
}
int igemm_halo_flat_launch(const omgsr_igemm_args& a, const IgemmGeo& g, hipStream_t st);       // igemm_halo_flat.hip
int igemm_halo_launch_mx(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    halo_geo(a, g, a.upsample != 0);
    if (g.flat) return igemm_halo_flat_launch(a, g, st);
    if (a.mx_fmt == 6) return igemm_halo_launch_mx6(a, g, st);
    const int rc = mx_attrs();
    if (rc != 0) return rc;
    if (a.upsample) hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 4, true>), g.interleave ? dim3(32 * ((g.ntm * g.ntn + 7) / 8)) : dim3(g.ntm * g.ntn, 4), dim3(256), LDS_BYTES, st, a, g);
    else hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, true>), dim3(g.ntm * g.ntn), dim3(256), LDS_BYTES, st, a, g);
    return (int)hipGetLastError();
}
}  // namespace omgsr
