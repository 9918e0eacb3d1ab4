"""HBM traffic per launch of the dominant kernel families from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: one
counter per pass, MI355X_MICROARCH.md §rocprofv3 PMC slots), corrected as MI355X_MICROARCH.md §HBM prescribes for
gfx950: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE tallies 128-B read requests at 64 B).
Usage: python tools/traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections, csv, json, re, sys

FAMILIES = [("igemm", re.compile(r"igemm_(halo_|dma_)?kernel|splitk_reduce")), ("groupnorm", re.compile(r"gn_(partial|apply|finalize)")),
            ("attention", re.compile(r"attn_kernel")), ("layernorm", re.compile(r"layernorm_kernel"))]


def collect(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for fam, rx in FAMILIES:
            if rx.search(r["Kernel_Name"]):
                agg[fam][0] += 1
                agg[fam][1] += float(r["Counter_Value"])
                break
    return agg


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 1 "
                 "--no-cpu-baseline --no-roofline; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md (gfx950)",
       "families": {}}
for fam in fetch:
    nf, f = fetch[fam]
    nw, w = write.get(fam, [0, 0.0])
    out["families"][fam] = {"launches_fetch_pass": nf, "launches_write_pass": nw,
                            "avg_fetch_kb_raw": f / max(nf, 1), "avg_write_kb": w / max(nw, 1),
                            "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["families"], indent=1))
