"""One process per GPU. The OMGSR hot path shards by image (images are independent: the reference
loops them serially, infer/infer_omgsr_s.py:69), so steady state needs NO collective. The only
exchange is one-time: rank 0 broadcasts the weights to its peers over RCCL/xGMI, followed by a
checksum all-reduce proving the replicas are bit-identical.

`backend="nccl"` is RCCL on ROCm; tests drive the same code with gloo on CPU (world_size 2).
"""
from __future__ import annotations

import datetime
import os
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist

BUCKET_BYTES = 512 << 20   # few, large collectives: xGMI links are per-peer, a broadcast is link-bound


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialises torch.distributed from the torchrun env (no-op for a single process)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        # explicit collective timeout, longer than any rank-0-only reporting leg of bench.py (the peers wait in the final barrier while
        # rank 0 runs its roofline pass): the default 10 min of the NCCL backend is not a contract
        kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("OMGSR_DIST_TIMEOUT_S", "1800")))}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def world_size_seen() -> int:
    """Ranks that actually answer on the process group: an all-reduce of 1 over RCCL (1 without a group). bench.py prints it as
    `n_ranks_seen` next to the launcher's WORLD_SIZE."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of `total` images over `world` ranks (first `total % world` ranks get one more)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _buckets(tensors: List[torch.Tensor], limit: int) -> Iterable[List[torch.Tensor]]:
    cur, size = [], 0
    for t in tensors:
        n = t.numel() * t.element_size()
        if cur and (size + n > limit or t.dtype != cur[0].dtype):
            yield cur
            cur, size = [], 0
        cur.append(t)
        size += n
    if cur:
        yield cur


@torch.no_grad()
def broadcast_module_(module: torch.nn.Module, src: int = 0, bucket_bytes: int = BUCKET_BYTES) -> int:
    """Broadcast every parameter and buffer from `src` in flat buckets; returns the bytes moved."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    tensors = [p.data for p in module.parameters()] + [b for b in module.buffers()]
    moved = 0
    for bucket in _buckets(tensors, bucket_bytes):
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.broadcast(flat, src=src)
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        moved += flat.numel() * flat.element_size()
    return moved


@torch.no_grad()
def module_checksum(module: torch.nn.Module) -> torch.Tensor:
    """Order-dependent 2-word checksum of all parameter BITS (int64 arithmetic, exact)."""
    dev = next(module.parameters()).device
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    for i, p in enumerate(module.parameters()):
        raw = p.data.contiguous().view(torch.int16 if p.element_size() == 2 else torch.int32).to(torch.int64)
        acc[0] += raw.sum()
        acc[1] += (raw * ((torch.arange(raw.numel(), device=dev, dtype=torch.int64).view_as(raw) % 8191) + 1)).sum() * (i + 1) % 1000003
    return acc


def replicas_identical(module: torch.nn.Module) -> bool:
    """all_reduce(MAX) and all_reduce(MIN) of the checksum agree <=> every rank holds the same bits."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True              # no peers: nothing to compare (and no 12 G-element reduction in front of a profiled run)
    cs = module_checksum(module)
    hi, lo = cs.clone(), cs.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return bool(torch.equal(hi, lo))


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown() -> None:
    """Every rank waits for the slowest one (rank 0 runs the untimed roofline leg after the timed region), then the process
    group is destroyed: no rank exits while a peer could still enter a collective, and RCCL tears down cleanly."""
    if dist.is_initialized():
        if dist.get_world_size() > 1:
            dist.barrier()
        dist.destroy_process_group()


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(x: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())
