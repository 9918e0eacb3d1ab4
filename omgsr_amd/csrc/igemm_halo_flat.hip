// FLAT instantiations of the halo-tile kernel body (igemm_halo_body.hip.h, FLAT = true): nine taps over the flattened padded map of a
// NARROW image (W <= 45: the tiled VAE's 1/8-resolution tile images), 256 consecutive positions per workgroup. A translation unit of its
// own: the run-time pitch costs the fragment-address table its compile-time geometry, and the spatial instantiations must not pay for it.
#include "igemm_halo_body.hip.h"

namespace omgsr {
static int flat_attrs() {
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(igemm_halo_kernel<bf16_t, 0, false, false, 9, false, true>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, false, true>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, true, true>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<bf16_t, false, 9, false, true>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, false, true>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, true, true>)};
        int rc = halo_set_lds_attr(fns, 6);
        if (rc != 0) return rc;
        // ... and with the 27-piece patch (maps 46-80 pixels wide: pitch > 47)
        const void* big[] = {reinterpret_cast<const void*>(igemm_halo_kernel<bf16_t, 0, false, false, 9, false, 2>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, false, 2>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, true, 2>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<bf16_t, false, 9, false, 2>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, false, 2>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, true, 2>)};
        rc = halo_set_lds_attr(big, 6, LDS_BYTES_BIG);
        if (rc != 0) return rc;
        attr_set = true;
    }
    return 0;
}
int igemm_halo_flat_launch_mx6(const omgsr_igemm_args& a, const IgemmGeo& g, hipStream_t st);        // igemm_halo_mx6_flat.hip
int igemm_halo_flat_launch_multi_mx6(const void* halo_multi, unsigned blocks, bool big, hipStream_t st);
// one problem per grid; g comes from halo_geo (g.flat != 0)
int igemm_halo_flat_launch(const omgsr_igemm_args& a, const IgemmGeo& g, hipStream_t st) {
    if (a.mx_chunks16 > 0 && a.mx_fmt == 6) return igemm_halo_flat_launch_mx6(a, g, st);
    const int rc = flat_attrs();
    if (rc != 0) return rc;
    const dim3 grid(g.ntm * g.ntn);
    if (g.flat > 47) {          // 256 + 2 P + 2 patch rows > 352: the 27-piece patch
        if (a.mx_chunks16 > 0) hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, true, 2>), grid, dim3(256), LDS_BYTES_BIG, st, a, g);
        else if (omgsr::compute_dtype() == 1) hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, false, 2>), grid, dim3(256), LDS_BYTES_BIG, st, a, g);
        else hipLaunchKernelGGL((igemm_halo_kernel<bf16_t, 0, false, false, 9, false, 2>), grid, dim3(256), LDS_BYTES_BIG, st, a, g);
        return (int)hipGetLastError();
    }
    if (a.mx_chunks16 > 0) hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, true, true>), grid, dim3(256), LDS_BYTES, st, a, g);
    else if (omgsr::compute_dtype() == 1) hipLaunchKernelGGL((igemm_halo_kernel<f16_t, 0, false, false, 9, false, true>), grid, dim3(256), LDS_BYTES, st, a, g);
    else hipLaunchKernelGGL((igemm_halo_kernel<bf16_t, 0, false, false, 9, false, true>), grid, dim3(256), LDS_BYTES, st, a, g);
    return (int)hipGetLastError();
}
// several problems per launch, at least one of them in the FLAT form (a problem with g.flat == 0 walks its spatial tiles in the same kernel)
int igemm_halo_flat_launch_multi(const void* halo_multi, unsigned blocks, hipStream_t st) {
    const HaloMulti& m = *reinterpret_cast<const HaloMulti*>(halo_multi);
    const int rc = flat_attrs();
    if (rc != 0) return rc;
    bool big = false;
    for (int i = 0; i < m.count; ++i) big = big || m.g[i].flat > 47;         // one kernel per launch: the 27-piece patch serves the narrower problems too
    if (m.p[0].mx_chunks16 > 0 && m.p[0].mx_fmt == 6) return igemm_halo_flat_launch_multi_mx6(halo_multi, blocks, big, st);
    if (big) {
        if (m.p[0].mx_chunks16 > 0) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, true, 2>), dim3(blocks), dim3(256), LDS_BYTES_BIG, st, m);
        else if (omgsr::compute_dtype() == 1) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, false, 2>), dim3(blocks), dim3(256), LDS_BYTES_BIG, st, m);
        else hipLaunchKernelGGL((igemm_halo_multi_kernel<bf16_t, false, 9, false, 2>), dim3(blocks), dim3(256), LDS_BYTES_BIG, st, m);
        return (int)hipGetLastError();
    }
    if (m.p[0].mx_chunks16 > 0) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, true, true>), dim3(blocks), dim3(256), LDS_BYTES, st, m);
    else if (omgsr::compute_dtype() == 1) hipLaunchKernelGGL((igemm_halo_multi_kernel<f16_t, false, 9, false, true>), dim3(blocks), dim3(256), LDS_BYTES, st, m);
    else hipLaunchKernelGGL((igemm_halo_multi_kernel<bf16_t, false, 9, false, true>), dim3(blocks), dim3(256), LDS_BYTES, st, m);
    return (int)hipGetLastError();
}
}  // namespace omgsr
