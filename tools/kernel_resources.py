"""Per-kernel register / spill / scratch / occupancy table of the library as built (CPU box, no GPU): parsed from hipcc's
-Rpass-analysis=kernel-resource-usage remarks, which omgsr_amd/build.py keeps next to every object.
    python tools/kernel_resources.py                 print the table
    python tools/kernel_resources.py --write         (re)write tests/golden/kernel_resources.json - the table
                                                     tests/test_kernel_resources_cpu.py pins the MFMA kernels against
Run --write after a deliberate kernel change, and look at the diff: a spill count or an occupancy that moved is a performance change."""
import json, os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from omgsr_amd.build import kernel_resources  # noqa: E402

PINNED_PREFIXES = ("igemm_", "attn_kernel", "splitk_reduce")
GOLDEN = os.path.join(HERE, "tests", "golden", "kernel_resources.json")


def pinned(table):
    keep = ("vgpr", "agpr", "sgpr", "spill_vgpr", "spill_sgpr", "scratch", "occupancy")
    return {k: {f: v[f] for f in keep if f in v} for k, v in sorted(table.items()) if k.startswith(PINNED_PREFIXES)}


if __name__ == "__main__":
    t = kernel_resources()
    if "--write" in sys.argv:
        json.dump(pinned(t), open(GOLDEN, "w"), indent=1)
        print(f"wrote {GOLDEN}: {len(pinned(t))} kernels")
    else:
        for k, v in sorted(t.items()):
            print(f"{v['source']:22s} {k:44s} V={v.get('vgpr'):>4} A={v.get('agpr'):>3} S={v.get('sgpr', 0):>3} spillV={v.get('spill_vgpr')} "
                  f"spillS={v.get('spill_sgpr')} scratch={v.get('scratch')} occ={v.get('occupancy')}")
