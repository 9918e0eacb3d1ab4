"""Host-side planning of the halo conv kernel's forms (no GPU: omgsr_igemm_multi_plan / omgsr_igemm_gn_slots only read the geometry).
A launch group runs ONE form: the plan marks the whole group FLAT (group_tiles < 0) when every problem can run it and the group then
needs fewer workgroup tiles, and the GroupNorm-statistics slot count a caller allocates follows the same decision (a mismatch between
the two would overrun the partials buffer). Geometry: omgsr_amd/csrc/igemm_halo_body.hip.h (halo_flat_eligible / halo_flat_pitch)."""
import ctypes as C

import pytest

from omgsr_amd import _lib


def _args(n, h, w, cin=512, cout=512):
    a = _lib.IgemmArgs()
    a.in_ = a.weight = a.weight_cm = a.out = 0x1000          # never dereferenced by the plan
    a.N, a.H, a.W, a.Cin, a.Cout, a.Cout_pad, a.K_pad = n, h, w, cin, cout, cout, 9 * cin
    a.R = a.S = 3
    a.stride, a.pad_top, a.pad_left, a.upsample = 1, 1, 1, 0
    a.Ho, a.Wo, a.batch, a.alpha = h, w, 1, 1.0
    a.gn_groups = 32
    return a


def _plan(shapes):
    lib = _lib.load()
    arr = (_lib.IgemmArgs * len(shapes))(*[_args(*s) for s in shapes])
    assert lib.omgsr_igemm_multi_plan(arr, len(shapes)) == 0
    return arr, [lib.omgsr_igemm_gn_slots(C.byref(arr[i])) for i in range(len(shapes))]


def _tiles(h, w, flat):
    return -(-h * (w + 2) // 256) if flat else -(-w // 32) * -(-h // 8)


@pytest.mark.parametrize("shapes,flat", [
    ([(36, 40, 40), (12, 40, 32), (12, 32, 40), (4, 32, 32)], True),      # tiled-VAE encoder, 1/8 level: 416 flat tiles against 532
    ([(36, 80, 80), (12, 80, 64), (12, 64, 80), (4, 64, 64)], True),      # 1/4 level: the 27-piece patch
    ([(4, 86, 86), (4, 86, 64), (4, 64, 86), (4, 64, 64)], False),        # decoder: 86 > 80 is not eligible, the group stays spatial
    ([(36, 32, 32), (12, 32, 32)], False),                                # exactly tiled already: flat would need more tiles
])
def test_one_form_per_launch_group(shapes, flat):
    arr, slots = _plan(shapes)
    ntn = 4                                                               # 512 output channels = four 128-column tiles
    total = sum(n * _tiles(h, w, flat) * ntn for n, h, w in shapes)
    for a, s, (n, h, w) in zip(arr, slots, shapes):
        assert a.group_tiles == (-total if flat else total)
        assert s == 2 * _tiles(h, w, flat)                                # two wave rows per tile, per image


def test_a_problem_on_its_own_decides_for_itself():
    lib = _lib.load()
    for (n, h, w), flat in [((64, 40, 40), True), ((64, 40, 32), False), ((64, 20, 20), True), ((16, 75, 75), True), ((16, 86, 86), False)]:
        a = _args(n, h, w)
        assert a.group_tiles == 0
        assert lib.omgsr_igemm_gn_slots(C.byref(a)) == 2 * _tiles(h, w, flat), (h, w)
    # a 16 x 16 map gains nothing from either form (two tiles each way, half of every spatial tile padding): it leaves the halo kernel, and its
    # statistics come in the GEMM-shaped kernels' layout, one slot per 32 output rows
    assert lib.omgsr_igemm_gn_slots(C.byref(_args(64, 16, 16))) == 16 * 16 // 32


# ---- round 5: the normalising patch producer (omgsr_igemm_gn_fusable) and the halo kernel's split-K plan (omgsr_igemm_workspace_bytes) --------

def test_groupnorm_fusion_policy():
    """GroupNorm apply as the conv's patch producer: only where the problem takes the halo-tile kernel's spatial nine-tap form with a plain
    16-bit operand and weight, the (scale, shift) table fits (Cin <= 1024) and the output is ONE 128-column tile (the measured break-even:
    profiles/r05_experiments.md). The answer never depends on whether the table is attached yet."""
    lib = _lib.load()

    def fus(n, h, w, cin, cout, **kw):
        a = _args(n, h, w, cin, cout)
        a.in_el = 0
        for k, v in kw.items():
            setattr(a, k, v)
        first = lib.omgsr_igemm_gn_fusable(C.byref(a))
        a.gn_scale_shift, a.gn_nimg, a.gn_act = 0x1000, 1, 1
        assert lib.omgsr_igemm_gn_fusable(C.byref(a)) == first
        return bool(first)
    assert fus(4, 320, 320, 128, 128)                    # tiled-VAE decoder, last level
    assert fus(4, 320, 320, 128, 8)                      # conv_out: the narrow shape
    assert not fus(4, 320, 320, 128, 256)                # two column tiles: break-even -> the apply pass stays
    assert not fus(4, 160, 160, 512, 512)
    assert not fus(64, 40, 40, 128, 128)                 # FLAT form (narrow map): no normalising instantiation
    assert not fus(1, 16, 16, 128, 128)                  # not a halo-kernel problem at all
    assert not fus(4, 320, 320, 2048, 128)               # table beyond the 8 KB the kernel holds
    assert not fus(4, 320, 320, 128, 128, in_el=1)       # fp32 stream (accurate tier): not built
    assert not fus(4, 320, 320, 256, 128, in_ld=128, w_split=1)      # wrapped contraction (weight split)
    assert not fus(4, 160, 160, 128, 128, upsample=1, Ho=320, Wo=320)


def _mx_args(n, h, w, c, cout, **kw):
    """3x3 conv over an OMGSR_EL_MX operand of c logical channels (Cin counts 16-bit slots)."""
    lib = _lib.load()
    lib.omgsr_set_compute_dtype(1)
    a = _args(n, h, w, 2 * c, cout)
    a.K_pad, a.mx_chunks16, a.sample_rows = 9 * 2 * c, c // 32, h * w
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_halo_split_k_plan():
    """One image per call (the reference's own operating point): the accurate tier's mixed-precision 3x3 convs are 16 ... 64 workgroup tiles;
    the halo-tile kernel then runs 2 ... 8 chunk ranges of the contraction as one launch group. The workspace the caller must hand over is
    splits x rows x padded columns fp32; big problems, launch-group members and wrapped contractions never split."""
    lib = _lib.load()
    try:
        def splits(a):
            cols = -(-a.Cout // 128) * 128
            b = lib.omgsr_igemm_workspace_bytes(C.byref(a))
            assert b % (4 * a.N * a.Ho * a.Wo * cols) == 0
            return b // (4 * a.N * a.Ho * a.Wo * cols)
        assert splits(_mx_args(1, 64, 64, 512, 512)) == 8          # 64 tiles -> 512 / 64 = 8 ranges: 4 fp16 + 4 fp8 (16 chunks each side)
        assert splits(_mx_args(1, 32, 32, 640, 640)) == 8          # 20 tiles: the cap
        assert splits(_mx_args(1, 64, 64, 320, 320)) == 8          # 48 tiles
        assert splits(_mx_args(1, 8, 64, 64, 128)) == 2            # two chunks per side: one fp16 range, one fp8 range
        assert splits(_mx_args(1, 9, 33, 64, 128)) == 0            # narrow map: the FLAT form, which has no chunk-range instantiation
        assert splits(_mx_args(8, 64, 64, 320, 320)) == 0          # batch 8: 384 tiles fill the chip, the one-pass epilogue keeps its fused statistics
        assert splits(_mx_args(1, 64, 64, 512, 512, group_tiles=4096)) == 0     # a member of a tiled-VAE launch group
        # batch-invariant mode decides from ONE sample: batch 8 then splits exactly like batch 1 (same summation order per element)
        lib.omgsr_set_batch_invariant(1)
        assert splits(_mx_args(8, 64, 64, 320, 320)) == 8
    finally:
        lib.omgsr_set_batch_invariant(0)
        lib.omgsr_set_compute_dtype(0)
