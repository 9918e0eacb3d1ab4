"""Per-kernel-family table of one rocprofv3 --pmc pass (csv): launches and the mean of every counter per launch.
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (1024 * shader cycles), shader cycles =
GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs), when both are present; LDS active % = SQ_LDS_IDX_ACTIVE / (256 CUs * cycles). Usage: pmc_table.py <counter_collection.csv>"""
import collections, csv, re, sys

FAM = [("igemm_halo_multi_kernel", re.compile(r"igemm_halo_multi_kernel")), ("igemm_halo_kernel", re.compile(r"igemm_halo_kernel")),
       ("igemm_p8_kernel", re.compile(r"igemm_p8_kernel")), ("igemm_gmx_kernel", re.compile(r"igemm_gmx_kernel")), ("igemm_dma_kernel", re.compile(r"igemm_dma_kernel")),
       ("igemm_kernel", re.compile(r"igemm_kernel")), ("splitk_reduce", re.compile(r"splitk_reduce")), ("attn_kernel", re.compile(r"attn_kernel")),
       ("gn_apply", re.compile(r"gn_apply")), ("layernorm", re.compile(r"layernorm"))]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    for fam, rx in FAM:
        if rx.search(r["Kernel_Name"]):
            a = agg[fam][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            break
counters = sorted({c for f in agg.values() for c in f})
print("| kernel | launches | " + " | ".join(counters) + " | MFMA busy % | LDS active % |")
print("|---|---|" + "---|" * (len(counters) + 2))
for fam, cs in agg.items():
    n = max(v[0] for v in cs.values())
    row = [f"{cs[c][1] / max(cs[c][0], 1):.3g}" if c in cs else "-" for c in counters]
    busy = "-"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs and cs["GRBM_GUI_ACTIVE"][1] > 0:
        busy = f"{100.0 * cs['SQ_VALU_MFMA_BUSY_CYCLES'][1] / (cs['GRBM_GUI_ACTIVE'][1] / 8 * 1024):.1f}"
    lds = "-"
    if "SQ_LDS_IDX_ACTIVE" in cs and "GRBM_GUI_ACTIVE" in cs and cs["GRBM_GUI_ACTIVE"][1] > 0:
        lds = f"{100.0 * cs['SQ_LDS_IDX_ACTIVE'][1] / (cs['GRBM_GUI_ACTIVE'][1] / 8 * 256):.1f}"
    print(f"| {fam} | {n} | " + " | ".join(row) + f" | {busy} | {lds} |")
