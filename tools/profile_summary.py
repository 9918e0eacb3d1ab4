"""Assemble profiles/<tag>_*.md + <tag>_traffic.json from what tools/profile_round.sh left under gpurun_out/prof_<dir>.
Usage: python tools/profile_summary.py gpurun_out/prof_r02b r02_s1024_accurate "title of the workload" [r02_traffic ["extra bench flags"]] """
import json, os, shutil, sys

src, tag, title = sys.argv[1], sys.argv[2], sys.argv[3]
ttag = sys.argv[4] if len(sys.argv) > 4 else tag + "_traffic"
extra = (" " + sys.argv[5]) if len(sys.argv) > 5 else ""
# the stdout line is compact since round 6; the full record (per-kernel rows) is the detail file written by the same run
_detail = os.path.join(src, "bench_detail.json")
bench = json.load(open(_detail)) if os.path.isfile(_detail) else json.loads(open(os.path.join(src, "bench_under_trace.json")).read().strip().splitlines()[-1])
traffic = json.load(open(os.path.join(src, "traffic.json")))
roof = bench["roofline"]
fam = traffic["families"]
lines = []
lines.append(f"# rocprofv3 --kernel-trace --stats of the default bench: {title}\n")
lines.append(f"Command (GPU box, `tools/profile_round.sh`): `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 "
             f"--no-cpu-baseline --no-fast-tiers --no-f1024 --no-latency{extra}`; config {json.dumps(bench['config'])}, dtype {bench['dtype']}. "
             f"5 pipeline passes in the trace (1 warm-up + 3 timed + 1 roofline pass).")
lines.append(f"bench line under the profiler: {bench['ms_per_step']} ms/step ({bench['value']} {bench['unit']}). bench.py's HIP-event leg in the "
             f"same run: igemm family {roof['kernel_ms']} ms / {roof['launches']} launches; per kernel (total ms per pass, algorithmic TFLOP/s): "
             + "; ".join(f"{k} {v['total_ms']} ms / {v['launches']} launches = {v['avg_us']} us avg, {v['achieved_tflops']} TF" for k, v in roof["kernels"].items()) + ".")
lines.append("The kernel-table rows x 1/5 reproduce those per-pass totals. Rows of torch's own kernels (`at::native`, `__amd_rocclr_*`) belong "
             "to model set-up (seeded init, weight packing, replica checksum) and are omitted below.\n")
lines.append(f"HBM traffic of the same command (`profiles/{ttag}.json`, separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, "
             f"hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024): " +
             "; ".join(f"{k} {v['hbm_bytes_per_launch'] / 1e6:.0f} MB per launch" for k, v in fam.items()) +
             f". Algorithmic bytes of the igemm family: {roof.get('algorithmic_bytes_per_launch', 0) / 1e6:.0f} MB per launch.\n")
# per-kernel traffic (round 4): measured HBM bytes per launch next to the algorithmic bytes bench.py counted for the same kernel, only where
# the PMC run saw exactly the launches of one pipeline pass that bench.py timed (tools/traffic_summary.py keeps the steady-state passes)
krows = traffic.get("kernels") or {}
if krows:
    lines.append("Per kernel (measured = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches of the steady-state passes; algorithmic = each operand, weight and output "
                 "byte once, as `omgsr_timing` counts them; `FETCH_SIZE` counts L2 misses, MALL hits included, so a ratio above 1 is re-reads that missed "
                 "the XCD's L2 - halo rows of neighbouring tiles, operands re-read by other N-tiles - not necessarily DRAM traffic):\n")
    lines.append("| kernel | launches / pass (PMC) | launches (bench) | measured MB / launch | algorithmic MB / launch | ratio |")
    lines.append("|---|---|---|---|---|---|")
    sk = roof["kernels"].get("igemm_dma_kernel(split-K)+splitk_reduce_kernel", {}).get("launches", 0)
    for k, v in krows.items():
        mine = roof["kernels"].get(k)
        n_b = (mine or {}).get("launches")
        if k == "igemm_dma_kernel" and sk:
            n_b = (n_b or 0) + sk
        alg = (mine or {}).get("bytes_per_launch")
        ratio = f"{v['hbm_bytes_per_launch'] / alg:.2f}" if (alg and not (k == 'igemm_dma_kernel' and sk)) else "-"
        lines.append(f"| {k} | {v['launches_per_pipeline_pass']} | {n_b if n_b is not None else '-'} | {v['hbm_bytes_per_launch'] / 1e6:.0f} | "
                     f"{alg / 1e6:.0f} |" .replace("None", "-") + f" {ratio} |" if alg else
                     f"| {k} | {v['launches_per_pipeline_pass']} | {n_b if n_b is not None else '-'} | {v['hbm_bytes_per_launch'] / 1e6:.0f} | - | - |")
    lines.append("")
lines.append("MFMA / LDS counters of the same command (`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT "
             "SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE`, `tools/pmc_table.py`; per-launch means):\n")
lines.append(open(os.path.join(src, "pmc_mfma.md")).read().rstrip() + "\n")
ks = open(os.path.join(src, "kernel_stats.md")).read().splitlines()
lines += [l for l in ks if not any(t in l for t in ("at::native", "__amd_rocclr", "elementwise_kernel_with_index", "hipcub", "rocprim"))]
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{tag}_kernel_stats.md", "w").write("\n".join(lines) + "\n")
traffic.update(bench.get("args", {}))            # bench.py matches the file to its own run by (workload, weight_dtype, batch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd.build import _source_digest        # noqa: E402
traffic["source_digest"] = _source_digest()      # bench.py reports whether the kernels changed since these PMC passes
json.dump(traffic, open(f"profiles/{ttag}.json", "w"), indent=1)
print(f"profiles/{tag}_kernel_stats.md", len(lines), "lines")
