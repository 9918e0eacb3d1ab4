// fp16 instantiations of the one-problem-per-grid halo-tile kernel (igemm_halo_body.hip.h); see igemm_halo.hip.
#include "igemm_halo_body.hip.h"

namespace omgsr {
int igemm_halo_launch_f16(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st, const bool phase, const bool narrow) {
    using T = f16_t;
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(igemm_halo_kernel<T, 0, false>), reinterpret_cast<const void*>(igemm_halo_kernel<T, 0, false, true>),
                             reinterpret_cast<const void*>(igemm_halo_kernel<T, 5, false>), reinterpret_cast<const void*>(igemm_halo_kernel<T, 0, false, false, 4>)};
        const int rc = halo_set_lds_attr(fns, (int)(sizeof(fns) / sizeof(fns[0])));
        if (rc != 0) return rc;
        attr_set = true;
    }
    const dim3 grid = (phase && g.interleave) ? dim3(32 * ((g.ntm * g.ntn + 7) / 8)) : dim3(g.ntm * g.ntn, phase ? 4 : 1, 1);
    static const char* var = getenv("OMGSR_HALO_VARIANT");     // A/B runs: "0" = LDS-DMA issued in front of the step's MFMAs
    if (phase) hipLaunchKernelGGL((igemm_halo_kernel<T, 0, false, false, 4>), grid, dim3(256), LDS_BYTES, st, a, g);
    else if (var && var[0] == '0' && !narrow) hipLaunchKernelGGL((igemm_halo_kernel<T, 5, false>), grid, dim3(256), LDS_BYTES, st, a, g);
    else if (narrow) hipLaunchKernelGGL((igemm_halo_kernel<T, 0, false, true>), grid, dim3(256), LDS_BYTES, st, a, g);
    else hipLaunchKernelGGL((igemm_halo_kernel<T, 0, false>), grid, dim3(256), LDS_BYTES, st, a, g);
    return (int)hipGetLastError();
}
}  // namespace omgsr
