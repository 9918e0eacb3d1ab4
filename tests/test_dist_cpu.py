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
