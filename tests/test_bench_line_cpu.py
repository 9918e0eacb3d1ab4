"""bench.py's stdout contract: ONE compact JSON line the driver's 8 KB stdout tail can hold, with `roofline` and
`cpu_baseline`; the full record goes to the detail file. CPU only: the line is assembled from a recorded full record."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _full_record():
    """Round 5's 23 KB default line (the one the driver could not parse) as the `full` record, per-kernel rows renamed to round 6's keys."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_line.json")))
    for row in full["roofline"]["kernels"].values():
        row["issued_tflops"], row["achieved_tflops"] = row.pop("achieved_tflops"), row.pop("handed_tflops")
        row["issued_frac"], row["frac"] = row.pop("frac"), round(row["achieved_tflops"] / 2500.0, 4)
    return full


def test_compact_line_fits_the_driver_tail_and_keeps_the_contract():
    b = _bench()
    full = _full_record()
    assert len(json.dumps(full)) > 20000
    line = b.compact_line(full, "S", 1024, 4, True, "fp32", "bench_detail.json", {})
    text = json.dumps(line)
    assert len(text) <= b.MAX_LINE_BYTES <= 6000, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert len(line["config"]["workload"]) <= 120 and "model" not in line["config"]
    rf = line["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    dom = next(iter(full["roofline"]["kernels"]))
    row = full["roofline"]["kernels"][dom]
    # the dominant kernel's ALGORITHMIC rate: handed FLOPs / its HIP-event time, not the issued MFMA work
    assert rf["kernel"] == dom and rf["achieved"] == row["achieved_tflops"] and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["achieved"] - row["handed_tflop"] / (row["total_ms"] * 1e-3)) / rf["achieved"] < 0.01
    assert rf["traffic_measured_in_this_run"] is False and rf["family"]["frac"] == full["roofline"]["frac"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 16 and cb["value"] > 0 and len(cb["sample"]) <= 160
    assert line["parity"]["rel_l2"] == full["parity"]["rel_l2"] and line["detail"] == "bench_detail.json"
    assert "f1024_bf16" in line["also"] and "ms_per_step" in line["also"]["f1024_bf16"]
    assert json.loads(text) == line


def test_compact_line_survives_failed_legs_and_oversize():
    b = _bench()
    full = _full_record()
    full.pop("workloads"); full.pop("other_tiers"); full["cpu_baseline"] = None; full["roofline"] = None; full["parity"] = None
    errors = {"f1024": "RuntimeError: " + "x" * 500, "cpu_baseline": "MemoryError: boom"}
    line = b.compact_line(full, "S", 1024, 4, True, "fp32", None, errors)
    assert line["roofline"] is None and line["cpu_baseline"] is None and line["value"] == full["value"]
    assert set(line["errors"]) == set(errors) and all(len(v) <= 160 for v in line["errors"].values())
    # a pathological record still cannot outgrow the tail: optional blocks are dropped first
    full = _full_record()
    full["other_tiers"] = {f"tier{i}": {"ms_per_step": 1.0, "images_per_s": 1.0, "rel_l2": 1e-4} for i in range(200)}
    line = b.compact_line(full, "S", 1024, 4, True, "fp32", "bench_detail.json", {})
    assert len(json.dumps(line)) <= b.MAX_LINE_BYTES and "roofline" in line and "cpu_baseline" in line
