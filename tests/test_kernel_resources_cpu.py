"""ISA pin of the hand-scheduled kernels: the register allocation, spill counts, scratch size and occupancy of
every MFMA kernel in the shipped library, as the compiler reports them at build time (-Rpass-analysis=kernel-resource-usage, kept by
omgsr_amd/build.py), against the committed table tests/golden/kernel_resources.json. The MX halo kernel lives at 256 registers with
13-14 spilled VGPRs and inline-asm MFMAs between hand-placed s_nops; a compiler or source change that re-spills it, or that drops a
kernel from two workgroups per CU to one, must fail HERE (build container) and not as a silent slowdown on the GPU box.
After a deliberate kernel change: `python tools/kernel_resources.py --write`, and read the diff."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tables():
    import shutil
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available: the resource table is read from the compiler's remarks at build time")
    if os.environ.get("OMGSR_BUILD_ABLATIONS", "0") == "1" or os.environ.get("OMGSR_EXTRA_DEFS"):
        pytest.skip("an ablation / variant build is not the shipped library the committed table describes")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import GOLDEN, pinned
    from omgsr_amd.build import kernel_resources
    return pinned(kernel_resources()), json.load(open(GOLDEN))


def test_every_pinned_kernel_is_still_built(tables):
    built, golden = tables
    missing = sorted(set(golden) - set(built))
    assert not missing, f"kernels in the committed table that the library no longer contains: {missing}"
    new = sorted(set(built) - set(golden))
    assert not new, f"MFMA kernels without a row in tests/golden/kernel_resources.json (run tools/kernel_resources.py --write): {new}"


def test_spills_scratch_and_occupancy_did_not_regress(tables):
    built, golden = tables
    bad = []
    for k, g in golden.items():
        b = built.get(k)
        if b is None:
            continue
        if b["occupancy"] < g["occupancy"]:
            bad.append(f"{k}: occupancy {g['occupancy']} -> {b['occupancy']} waves/SIMD")
        if b["spill_vgpr"] > g["spill_vgpr"]:
            bad.append(f"{k}: spilled VGPRs {g['spill_vgpr']} -> {b['spill_vgpr']}")
        if b["scratch"] > g["scratch"]:
            bad.append(f"{k}: scratch {g['scratch']} -> {b['scratch']} bytes/lane")
        if b["vgpr"] + b.get("agpr", 0) > 256 and b["occupancy"] >= 2:
            bad.append(f"{k}: {b['vgpr']} + {b.get('agpr', 0)} registers cannot give 2 waves/SIMD")
    assert not bad, "kernel resources regressed against tests/golden/kernel_resources.json:\n  " + "\n  ".join(bad)


def test_the_tight_kernels_are_where_design_says(tables):
    """the accumulator-heavy kernels run two workgroups per CU (<= 256 registers), the MX halo kernel's spills stay in the
    low teens, nothing else that ships in the hot path spills more than a handful."""
    built, _ = tables
    for k, b in built.items():
        if k.startswith(("igemm_halo", "igemm_dma", "igemm_p8", "igemm_gmx", "attn_kernel")):
            assert b["occupancy"] >= 2, (k, b)
    def targs(k):
        """{TAPS, MX, FLAT, GNF, SPLITK, PRIO, OUT6} of a halo instantiation: igemm_halo_kernel<T, ABL, PRIO, NARROW, TAPS, MX, FLAT, GNF, OUT6>,
        igemm_halo_multi_kernel<T, NARROW, TAPS, MX, FLAT, GNF, SPLITK, OUT6>."""
        a = k[k.index("<") + 1:-1].split(",")
        if k.startswith("igemm_halo_multi_kernel"):
            return dict(taps=a[2], mx=a[3], flat=a[4], gnf=a[5], splitk=a[6], out6=a[7], prio="0", abl="0")
        return dict(taps=a[4], mx=a[5], flat=a[6], gnf=a[7], out6=a[8], splitk="0", prio=a[2], abl=a[1])
    halo = {k: (b, targs(k)) for k, b in built.items() if k.startswith("igemm_halo")}
    mx = [b for b, t in halo.values() if t["taps"] == "9" and t["mx"] == "1" and t["splitk"] == "0" and t["out6"] == "0"]      # nine-tap MX instantiations, spatial and FLAT form
    assert len(mx) == 6 and all(b["spill_vgpr"] <= 16 and b["scratch"] <= 64 for b in mx), mx
    flat = [b for b, t in halo.values() if t["flat"] in ("1", "2")]      # the FLAT form (22- and 27-piece patch): 9 fragment-address registers instead of 36
    assert len(flat) == 16 and all(b["spill_vgpr"] <= 4 for b in flat), flat
    sk = [b for b, t in halo.values() if t["splitk"] == "1"]             # the split-K (chunk range) instantiations: spatial MX forms only, small-M regime
    assert len(sk) == 2 and all(b["spill_vgpr"] <= 24 and b["occupancy"] >= 2 for b in sk), sk
    mx6 = [b for b, t in halo.values() if t["mx"] == "6"]                # fp6 correction chunks (round 5): 24-byte fragments + a scale byte, 3 spilled registers
    assert len(mx6) == 11 and all(b["spill_vgpr"] <= 4 and b["scratch"] <= 16 and b["occupancy"] >= 2 for b in mx6), mx6
    out6 = [b for b, t in halo.values() if t["out6"] == "1"]             # the epilogue writes the fp6 operand form: a template flag with four instantiations of its own
    assert len(out6) == 4 and all(b["spill_vgpr"] <= 4 and b["scratch"] <= 16 and b["occupancy"] >= 2 for b in out6), out6
    gn = [b for b, t in halo.values() if t["gnf"] == "1"]                # GroupNorm apply as the patch producer (round 5): no spills, two workgroups per CU
    assert len(gn) == 8 and all(b["spill_vgpr"] == 0 and b["scratch"] == 0 for b in gn), gn
    for k, (b, t) in halo.items():
        if not (t["taps"] == "9" and t["mx"] in ("1", "6") and t["flat"] == "0") and t["flat"] == "0" and t["prio"] == "0":      # (the s_setprio A/B instantiation spills a few)
            assert b["spill_vgpr"] == 0, (k, b)
