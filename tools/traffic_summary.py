"""HBM traffic per launch, ONE ROW PER KERNEL NAME, from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: one counter per pass,
MI355X_MICROARCH.md §rocprofv3 PMC slots), corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950:
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE tallies 128-B read requests at 64 B).

Round 3's version matched `igemm_(halo_|dma_)?kernel` and silently dropped igemm_halo_multi_kernel / igemm_p8_kernel - the
dominant kernels of S-1024 / F-1024. Now every kernel of the library is classified by its demangled template
name into the SAME names bench.py's `roofline.kernels` uses (`igemm_halo_multi_kernel`, `igemm_halo_multi_kernel<TAPS=4>`,
`igemm_halo_kernel`, `igemm_halo_kernel<TAPS=4>`, `igemm_p8_kernel`, `igemm_dma_kernel`, `igemm_kernel`, `splitk_reduce_kernel`,
`attn_kernel`, ...), launch counts are kept per pipeline pass, and bench.py only quotes a kernel's traffic when the count equals
the launches it timed itself. `families` (igemm = every igemm kernel) is kept for the whole-family figure.

Only the steady-state passes at the end of the run are counted (see steady_rows).
Usage: python tools/traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [passes if no period is found = 2]"""
import collections, csv, json, re, sys

# (bench name, regex over the rocprofv3 Kernel_Name) - first match wins; mangled names: igemm_halo_multi_kernelI<T>Lb<NARROW>ELi<TAPS>ELb<MX>E,
# igemm_halo_kernelI<T>Li<ABL>ELb<PRIO>ELb<NARROW>ELi<TAPS>ELb<MX>E
KERNELS = [
    # round 5: ... ELi<FLAT>ELi<GNF>ELb<SPLITK>E follow; GNF = 1 are the GroupNorm-fused instantiations (bench: `<GN>`); MX is an int since the
    # fp6 form (L[bi]: older traces carry the bool mangling): MX = 6 are the fp6 instantiations (bench: `<fp6>`)
    ("igemm_halo_multi_kernel<GN>", re.compile(r"igemm_halo_multi_kernelI\w+?_?Lb[01]ELi9EL[bi]0ELi0ELi1E")),
    ("igemm_halo_kernel<GN>", re.compile(r"igemm_halo_kernelI\w+?_?Li\dELb[01]ELb[01]ELi9EL[bi]0ELi0ELi1E")),
    ("igemm_halo_multi_kernel<fp6>", re.compile(r"igemm_halo_multi_kernelI\w+?_?Lb[01]ELi9ELi6E")),
    ("igemm_halo_kernel<fp6>", re.compile(r"igemm_halo_kernelI\w+?_?Li\dELb[01]ELb[01]ELi9ELi6E")),
    ("igemm_halo_multi_kernel<TAPS=4>", re.compile(r"igemm_halo_multi_kernelI\w+?_?Lb[01]ELi4E|igemm_halo_multi_kernel<[^>]*, 4,")),
    ("igemm_halo_multi_kernel", re.compile(r"igemm_halo_multi_kernel")),
    ("igemm_halo_kernel<TAPS=4>", re.compile(r"igemm_halo_kernelI\w+?_?Li\dELb[01]ELb[01]ELi4E|igemm_halo_kernel<[^>]*, 4,")),
    ("igemm_halo_kernel", re.compile(r"igemm_halo_kernel")),
    ("igemm_p8_kernel", re.compile(r"igemm_p8_kernel")),
    ("igemm_gmx_kernel", re.compile(r"igemm_gmx_kernel")),
    ("igemm_dma_kernel", re.compile(r"igemm_dma_kernel")),
    ("igemm_kernel", re.compile(r"igemm_kernel")),
    ("splitk_reduce_kernel", re.compile(r"splitk_reduce")),
    ("attn_kernel", re.compile(r"attn_kernel")),
    ("gn_apply", re.compile(r"gn_apply")), ("gn_partial", re.compile(r"gn_partial")), ("gn_finalize", re.compile(r"gn_finalize")),
    ("layernorm_kernel", re.compile(r"layernorm")), ("softmax_rows_kernel", re.compile(r"softmax_rows")),
    ("rmsnorm_rope_kernel", re.compile(r"rmsnorm_rope")), ("to_operand_kernel", re.compile(r"to_operand")),
]
FAMILY = {"igemm": re.compile(r"^(igemm_|splitk_reduce)"), "groupnorm": re.compile(r"^gn_"), "attention": re.compile(r"^attn_kernel"),
          "layernorm": re.compile(r"^layernorm")}


MIN_PERIOD = 200


def classify(kname):
    for name, rx in KERNELS:
        if rx.search(kname):
            return name
    return None


def steady_rows(path, counter):
    """Rows of `counter` in dispatch order, cut down to the steady-state pipeline passes at the END of the run: the first pass of a
    process also launches one-time constant folding (cross-attention K / V, modulation tables ...), so 'launches / passes' over the
    whole file is not a whole number. The kernel-name sequence of steady passes is periodic: find the smallest period P with
    names[-P:] == names[-2P:-P] and keep the last 2 P dispatches (2 passes). Returns (rows, passes, period or None)."""
    rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort()
    names = [r[1] for r in rows]
    n = len(names)
    for P in range(MIN_PERIOD, n // 2 + 1):        # (a pipeline pass is thousands of dispatches; short repeats inside one are not passes)
        if names[n - P:] == names[n - 2 * P:n - P]:
            return rows[n - 2 * P:], 2, P
    return rows, None, None


def collect(path, counter):
    rows, passes, period = steady_rows(path, counter)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for _, kname, val in rows:
        name = classify(kname)
        if name:
            agg[name][0] += 1
            agg[name][1] += val
    return agg, passes, period


def main():
    (fetch, pf, period_f), (write, pw, period_w) = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    passes = pf if (pf and pf == pw and period_f == period_w) else (int(sys.argv[4]) if len(sys.argv) > 4 else 2)
    out = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline --no-roofline; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md (gfx950); "
                     "one row per kernel name, launches per pipeline pass = launches in the PMC run / passes",
           "passes": passes, "steady_state_period_dispatches": period_f if period_f == period_w else None, "kernels": {}, "families": {}}
    for name in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(name, [0, 0.0])
        nw, w = write.get(name, [0, 0.0])
        out["kernels"][name] = {"launches_fetch_pass": nf, "launches_write_pass": nw,
                                "launches_per_pipeline_pass": nf / passes if nf == nw else None,
                                "avg_fetch_kb_raw": f / max(nf, 1), "avg_write_kb": w / max(nw, 1),
                                "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
                                "hbm_bytes_total_per_pipeline_pass": (2.0 * f + w) * 1024.0 / passes}
    for fam, rx in FAMILY.items():
        rows = [(k, v) for k, v in out["kernels"].items() if rx.search(k)]
        if not rows:
            continue
        nf = sum(v["launches_fetch_pass"] for _, v in rows)
        nw = sum(v["launches_write_pass"] for _, v in rows)
        f = sum(v["avg_fetch_kb_raw"] * v["launches_fetch_pass"] for _, v in rows)
        w = sum(v["avg_write_kb"] * v["launches_write_pass"] for _, v in rows)
        out["families"][fam] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "kernels": [k for k, _ in rows],
                                "avg_fetch_kb_raw": f / max(nf, 1), "avg_write_kb": w / max(nw, 1),
                                "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: (v["launches_per_pipeline_pass"], round(v["hbm_bytes_per_launch"] / 1e6, 1)) for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
