"""One-GPU smoke of the RCCL side of omgsr_amd.dist (GPU box): the 8-GPU node the scaling bench needs has never been available, so this is the
only part of the N > 1 path that can touch hardware - process-group creation with backend "nccl" (= RCCL on ROCm) bound to cuda:0 with the
explicit timeout, a broadcast and the checksum / count all-reduces bench.py issues, on a group of ONE rank. It proves the library loads, the
device binding and the collectives' call signatures; it says nothing about xGMI.    python tools/rccl_selftest.py"""
import datetime
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import dist as D      # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
t = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
dist.broadcast(t, src=0)
one = torch.ones(1, dtype=torch.int64, device="cuda")
dist.all_reduce(one, op=dist.ReduceOp.SUM)
m = torch.nn.Linear(64, 64).cuda()
print("backend", dist.get_backend(), "world", dist.get_world_size(), "ranks seen", D.world_size_seen(), "count", int(one.item()),
      "replicas identical", D.replicas_identical(m), "max_over_ranks", D.max_over_ranks(1.5, torch.device("cuda", 0)))
D.barrier()
D.shutdown()
print("RCCL_SELFTEST_OK")
