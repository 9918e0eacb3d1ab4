"""N > 1 path on CPU: world_size-2 gloo processes exercise sharding, the weight broadcast and the
replica checksum (the same code runs over RCCL/xGMI on the GPU node)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from omgsr_amd.dist import shard_range
    for total in (1, 7, 8, 64, 65):
        for world in (1, 2, 4, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_range(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from omgsr_amd import dist as D
    from omgsr_amd.diffusers_api import AutoencoderKL
    from omgsr_amd.testing import seeded_init_
    r, _, w = D.init("gloo")
    vae = AutoencoderKL(block_out_channels=[32, 32, 64, 64], layers_per_block=1)
    if r == 0:
        seeded_init_(vae, 5)
    vae = vae.to(torch.bfloat16)
    same_before = D.replicas_identical(vae)
    moved = D.broadcast_module_(vae, src=0, bucket_bytes=1 << 16)      # small buckets: many collectives
    same_after = D.replicas_identical(vae)
    ref = seeded_init_(AutoencoderKL(block_out_channels=[32, 32, 64, 64], layers_per_block=1), 5).to(torch.bfloat16)
    equal = all(torch.equal(a, b) for a, b in zip(vae.state_dict().values(), ref.state_dict().values()))
    t = D.max_over_ranks(float(r + 1), torch.device("cpu"))
    D.barrier()
    q.put((r, w, same_before, moved, same_after, equal, t))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_broadcast_and_checksum_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    nbytes = sum(v.numel() * 2 for v in __import__("omgsr_amd.diffusers_api", fromlist=["x"]).AutoencoderKL(
        block_out_channels=[32, 32, 64, 64], layers_per_block=1).state_dict().values())
    for r, w, same_before, moved, same_after, equal, t in res:
        assert w == 2 and not same_before and same_after and equal
        assert moved == nbytes and t == 2.0


def _tiny_flux(device=None):
    from omgsr_amd.diffusers_api import FluxTransformer2DModel
    kw = dict(num_layers=2, num_single_layers=3, attention_head_dim=128, num_attention_heads=2, joint_attention_dim=64,
              pooled_projection_dim=32, in_channels=64)
    if device is None:
        return FluxTransformer2DModel(**kw)
    with torch.device(device):
        return FluxTransformer2DModel(**kw)


def _mixed_dtypes_(m):
    """bf16 weights with fp32 norm tables and biases in between: every bucket boundary of the broadcast that falls on a dtype change."""
    m.to(torch.bfloat16)
    for name, p in m.named_parameters():
        if "norm" in name or name.endswith(".bias"):
            p.data = p.data.float()
    return m


def _worker8(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from omgsr_amd import dist as D
    from omgsr_amd.testing import seeded_init_
    r, _, w = D.init("gloo")
    if r == 0:
        m = _mixed_dtypes_(seeded_init_(_tiny_flux(), 11, rounded=False))
    else:                                                      # peers never initialise weights: meta -> to_empty (uninitialised memory), as on the GPU node
        m = _mixed_dtypes_(_tiny_flux("meta")).to_empty(device="cpu")
    same_before = D.replicas_identical(m)
    moved = D.broadcast_module_(m, src=0, bucket_bytes=1 << 20)          # ~25 MB of weights: dozens of buckets, several dtype boundaries
    same_after = D.replicas_identical(m)
    ref = _mixed_dtypes_(seeded_init_(_tiny_flux(), 11, rounded=False))
    equal = all(a.dtype == b.dtype and torch.equal(a, b) for a, b in zip(m.state_dict().values(), ref.state_dict().values()))
    lo, hi = D.shard_range(64, r, w)
    ms = 100.0 + r
    q.put((r, w, same_before, moved, same_after, equal, (lo, hi), D.min_over_ranks(ms, torch.device("cpu")), D.max_over_ranks(ms, torch.device("cpu")),
           D.world_size_seen()))
    D.shutdown()


@pytest.mark.timeout(420)
def test_world8_flux_shaped_broadcast_and_shards():
    """BASELINE configs[4] on CPU: 8 gloo ranks, 64 images sharded 8 x 8, a FLUX-shaped module built on the meta device on
    the peers (to_empty) and filled by the bucketed broadcast across fp32 / bf16 boundaries, bit-exact replica checksum, and the per-rank
    min / max that bench.py prints next to the max-over-ranks time."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = _mixed_dtypes_(_tiny_flux())
    nbytes = sum(p.numel() * p.element_size() for p in ref.parameters()) + sum(b.numel() * b.element_size() for b in ref.buffers())
    assert len({p.dtype for p in ref.parameters()}) == 2
    for r, w, same_before, moved, same_after, equal, shard, tmin, tmax, seen in res:
        assert w == 8 and seen == 8 and not same_before and same_after and equal
        assert moved == nbytes and shard == (8 * r, 8 * r + 8) and (tmin, tmax) == (100.0, 107.0)


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` outside a torchrun environment launches 2 ranks itself (child torchrun, parent never touches a
    GPU) and rank 0 reports n_gpus 2, dp2 and the bytes the weight broadcast moved. CPU rehearsal: gloo, reduced models."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run-cpu"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["world_size"] == 2 and j["config"]["parallelism"].startswith("dp2")
    assert j["config"]["broadcast_bytes"] > 1 << 20 and j["config"]["global_batch"] == 4 and j["scaling"] == "weak"
    # per-rank step time next to the max over ranks the contract times: a SCALE run shows stragglers
    pr = j["per_rank_ms_per_step"]
    assert pr["min"] <= pr["max"] and abs(pr["max"] - j["ms_per_step"]) < 1e-6


def test_bench_dry_run_world8_line_is_compact(tmp_path):
    """configs[4]'s launch shape (`python bench.py --gpus 8 --workload f1024` on the GPU node) rehearsed on CPU: 8 ranks spawned by bench.py
    itself, gloo, reduced models; the ONE stdout line must fit the driver's stdout tail and report all 8 ranks."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "0", "--dry-run-cpu", "--batch", "8"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 6000, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["n_ranks_seen"] == 8 and j["config"]["global_batch"] == 64 and j["config"]["images_rank0"] == [0, 8]
    assert j["scaling"] == "weak" and j["dry_run"] is True and j["config"]["broadcast_bytes"] > 1 << 20
