// GNF instantiations of the halo-tile kernel (igemm_halo_body.hip.h): GroupNorm apply (+ SiLU) as the conv's patch producer. Their own
// translation unit: the in-place transform adds ~60 VALU instructions per piece to a loop that lives at 244 registers, and hipcc compiles
// the files in parallel.
#include "igemm_halo_body.hip.h"

namespace omgsr {
static int gn_attrs() {
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(igemm_halo_kernel<bf16_t, 0, false, false, 9, false, 0, 1>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, false, 9, false, 0, 1>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<bf16_t, 0, false, true, 9, false, 0, 1>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<f16_t, 0, false, true, 9, false, 0, 1>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<bf16_t, false, 9, false, 0, 1>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9, false, 0, 1>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<bf16_t, true, 9, false, 0, 1>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, true, 9, false, 0, 1>)};
        const int rc = halo_set_lds_attr(fns, (int)(sizeof(fns) / sizeof(fns[0])), LDS_BYTES_GN);
        if (rc != 0) return rc;
        attr_set = true;
    }
    return 0;
}

// what the fused form can run (the dispatcher adds the halo kernel's own preconditions and tile-count policy: igemm.hip)
bool igemm_halo_gn_geometry_ok(const omgsr_igemm_args& a) {
    return a.in_el == OMGSR_EL_16 && a.R == 3 && a.S == 3 && a.stride == 1 && !a.upsample && a.in_ld == 0 && !a.in_split && !a.w_split && a.mx_chunks16 == 0 &&
           (a.Cin % 32) == 0 && a.Cin <= GN_MAX_CIN && a.batch == 1 && a.weight_cm != nullptr;
}

int igemm_halo_gn_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    const bool narrow = halo_geo(a, g, false);
    if (g.flat || a.gn_nimg <= 0) return OMGSR_E_SHAPE;
    const int rc = gn_attrs();
    if (rc != 0) return rc;
    const dim3 grid(g.ntm * g.ntn);
    if (narrow) OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_halo_kernel<T, 0, false, true, 9, false, 0, 1>), grid, dim3(256), LDS_BYTES_GN, st, a, g));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_halo_kernel<T, 0, false, false, 9, false, 0, 1>), grid, dim3(256), LDS_BYTES_GN, st, a, g));
    return (int)hipGetLastError();
}

int igemm_halo_gn_launch_multi(const void* halo_multi, const unsigned blocks, const bool narrow, hipStream_t st) {
    const HaloMulti& m = *reinterpret_cast<const HaloMulti*>(halo_multi);
    const int rc = gn_attrs();
    if (rc != 0) return rc;
    if (narrow) OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_halo_multi_kernel<T, true, 9, false, 0, 1>), dim3(blocks), dim3(256), LDS_BYTES_GN, st, m));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_halo_multi_kernel<T, false, 9, false, 0, 1>), dim3(blocks), dim3(256), LDS_BYTES_GN, st, m));
    return (int)hipGetLastError();
}
}  // namespace omgsr
