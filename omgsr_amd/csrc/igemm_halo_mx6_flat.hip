// fp6 instantiations of the halo-tile kernel body in its FLAT forms (narrow maps: the tiled VAE's tile images at 1/4 and 1/8 resolution);
// see igemm_halo_mx6.hip for the format and igemm_halo_flat.hip for the form.
#include "igemm_halo_body.hip.h"

namespace omgsr {
static int mx6_flat_attrs() {
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, 6, 1>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, 6, 1>)};
        int rc = halo_set_lds_attr(fns, 2);
        if (rc != 0) return rc;
        const void* big[] = {reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, 6, 2>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, 6, 2>)};
        rc = halo_set_lds_attr(big, 2, LDS_BYTES_BIG);
        if (rc != 0) return rc;
        attr_set = true;
    }
    return 0;
}
int igemm_halo_flat_launch_mx6(const omgsr_igemm_args& a, const IgemmGeo& g, hipStream_t st) {
    const int rc = mx6_flat_attrs();
    if (rc != 0) return rc;
    const dim3 grid(g.ntm * g.ntn);
    if (g.flat > 47) hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, 6, 2>), grid, dim3(256), LDS_BYTES_BIG, st, a, g);
    else hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, 6, 1>), grid, dim3(256), LDS_BYTES, st, a, g);
    return (int)hipGetLastError();
}
int igemm_halo_flat_launch_multi_mx6(const void* halo_multi, unsigned blocks, bool big, hipStream_t st) {
    const HaloMulti& m = *reinterpret_cast<const HaloMulti*>(halo_multi);
    const int rc = mx6_flat_attrs();
    if (rc != 0) return rc;
    if (big) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, 6, 2>), dim3(blocks), dim3(256), LDS_BYTES_BIG, st, m);
    else hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, 6, 1>), dim3(blocks), dim3(256), LDS_BYTES, st, m);
    return (int)hipGetLastError();
}
}  // namespace omgsr
