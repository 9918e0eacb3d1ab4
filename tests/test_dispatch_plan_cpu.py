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
