// The halo-tile kernel body (igemm_halo_body.hip.h) as (1) igemm_halo_multi_kernel: several problems that share weights and epilogue
// options in ONE launch (the tiled VAE's tile-shape groups: corner / edge / interior tiles are separate dense tensors); (2) the
// mixed-precision instantiations (fp16 chunks + block-scaled fp8 chunks of an OMGSR_EL_MX operand) are in igemm_halo_mx.hip.
#include "igemm_halo_body.hip.h"

namespace omgsr {
static int multi_attrs() {
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {reinterpret_cast<const void*>(igemm_halo_multi_kernel<bf16_t, false, 9>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 9>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<bf16_t, true, 9>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, true, 9>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<bf16_t, false, 4>),
                             reinterpret_cast<const void*>(igemm_halo_multi_kernel<f16_t, false, 4>)};
        const int rc = halo_set_lds_attr(fns, (int)(sizeof(fns) / sizeof(fns[0])));
        if (rc != 0) return rc;
        attr_set = true;
    }
    return 0;
}

int igemm_halo_launch_multi_mx(const void* halo_multi, unsigned blocks, hipStream_t st);
int igemm_halo_gn_launch_multi(const void* halo_multi, unsigned blocks, bool narrow, hipStream_t st);      // igemm_halo_gn.hip
int igemm_halo_out6_launch_multi(const void* halo_multi, unsigned blocks, hipStream_t st);                 // igemm_halo_out6.hip
int igemm_halo_flat_launch_multi(const void* halo_multi, unsigned blocks, hipStream_t st);      // igemm_halo_flat.hip       // igemm_halo_mx.hip (HaloMulti has no linkage: same header, opaque pointer)

// `count` problems (<= HALO_MULTI_MAX) in one launch; all of them take the same kernel shape (the caller checks: same weights,
// Cin / Cout, epilogue options; `phase`, narrow-ness and the mixed-precision form therefore agree). The workgroup looks its problem up
// in a prefix table of tile counts; each problem's range starts on a multiple of 8 blocks so the XCD remap of its local index keeps
// its meaning (the filler blocks exit).
int igemm_halo_launch_multi(const omgsr_igemm_args* a, const IgemmGeo* g0, const int count, hipStream_t st, const bool phase) {
    HaloMulti m;
    m.count = count;
    int at = 0;
    bool narrow = false, flat = false;
    for (int i = 0; i < count; ++i) {
        m.p[i] = a[i];
        m.g[i] = g0[i];
        narrow = halo_geo(a[i], m.g[i], phase);
        flat = flat || m.g[i].flat != 0;                 // (omgsr_igemm_multi groups problems by form: all of a launch's problems agree)
        m.start[i] = at;
        at += (m.g[i].ntm * m.g[i].ntn + 7) & ~7;
    }
    m.start[count] = at;
    const int rc = multi_attrs();
    if (rc != 0) return rc;
    const dim3 grid = (phase && m.g[0].interleave) ? dim3(4 * at) : dim3(at, phase ? 4 : 1, 1);
    if (m.g[0].cc1 > 0 && (flat || phase || narrow || a[0].mx_chunks16 <= 0 || a[0].gn_scale_shift)) return OMGSR_E_SHAPE;      // split-K: the spatial MX instantiation only (halo_splitk_plan)
    if (a[0].out_mx == 6) return (phase || flat || narrow || m.g[0].cc1 > 0) ? OMGSR_E_SHAPE : igemm_halo_out6_launch_multi(&m, (unsigned)at, st);
    if (a[0].gn_scale_shift) return (phase || flat) ? OMGSR_E_SHAPE : igemm_halo_gn_launch_multi(&m, (unsigned)at, narrow, st);
    if (flat) return igemm_halo_flat_launch_multi(&m, (unsigned)at, st);
    if (a[0].mx_chunks16 > 0) return igemm_halo_launch_multi_mx(&m, (unsigned)at, st);
    if (phase) OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_halo_multi_kernel<T, false, 4>), grid, dim3(256), LDS_BYTES, st, m));
    else if (narrow) OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_halo_multi_kernel<T, true, 9>), grid, dim3(256), LDS_BYTES, st, m));
    else OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_halo_multi_kernel<T, false, 9>), grid, dim3(256), LDS_BYTES, st, m));
    return (int)hipGetLastError();
}

// Split-K (igemm.hip: halo_splitk_plan): `splits` chunk ranges of ONE problem as the members of a launch group. Each member is the problem itself with
// its epilogue stripped (fp32 partial tile into its slice of the workspace, no bias / activation / gate / residual / statistics: splitk_reduce_kernel
// applies them once to the sum) and its chunk range in the geometry block.
int igemm_halo_launch_splitk(const omgsr_igemm_args& a, const IgemmGeo& g0, const int splits, hipStream_t st) {
    if (splits < 2 || splits > HALO_MULTI_MAX || !a.workspace) return OMGSR_E_BADARG;
    omgsr_igemm_args parts[HALO_MULTI_MAX];
    IgemmGeo geos[HALO_MULTI_MAX];
    const int nk = a.Cin / 32;
    const bool mx = a.mx_chunks16 > 0;
    const int n16 = mx ? a.mx_chunks16 : nk, n8 = nk - n16;
    const int64_t M = (int64_t)a.N * a.Ho * a.Wo;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    const int ldw = ((logical_cols + 127) / 128) * 128;
    for (int s = 0; s < splits; ++s) {
        omgsr_igemm_args& q = parts[s];
        q = a;
        q.out = reinterpret_cast<float*>(a.workspace) + (int64_t)s * M * ldw;
        q.out_dtype = OMGSR_OUT_F32; q.out_ld = ldw; q.out_lo_off = 0; q.out_mx = 0;
        q.bias = nullptr; q.gate = nullptr; q.residual = nullptr; q.act = OMGSR_ACT_NONE; q.alpha = 1.0f;
        q.gn_partial = nullptr; q.gn_groups = 0; q.gn_entries = 0; q.overflow_flag = nullptr; q.workspace = nullptr;
        // halo_splitk_plan decided on the REAL arguments that this problem runs the spatial form (an out_mx = 6 problem is never FLAT-eligible, so it
        // plans a split on maps <= 80 wide too); the stripped view (out_mx = 0) WOULD be FLAT-eligible there and halo_geo() would pick FLAT, for which
        // no chunk-range instantiation exists: a positive group_tiles pins every part to the spatial form (halo_flat_pitch).
        q.group_tiles = g0.ntm > 0 && g0.ntn > 0 ? g0.ntm * g0.ntn : 1;
        geos[s] = g0;
        if (mx) {
            const int h = splits / 2, t = s < h ? s : s - h;
            geos[s].cc0 = s < h ? (n16 * t) / h : n16 + (n8 * t) / h;
            geos[s].cc1 = s < h ? (n16 * (t + 1)) / h : n16 + (n8 * (t + 1)) / h;
        } else {
            geos[s].cc0 = (nk * s) / splits;
            geos[s].cc1 = (nk * (s + 1)) / splits;
        }
    }
    return igemm_halo_launch_multi(parts, geos, splits, st, false);
}

}  // namespace omgsr
