"""Multi-GPU plumbing: one process per GPU, batch split per device, NO data-path collective.

ML-KEM / ML-DSA batch items are independent, so ranks never exchange data; the
process group (RCCL on GPUs, gloo in the CPU tests) is used only for the timing barrier and for
reducing per-rank timings / counters to rank 0.
"""
import os

import torch


def shard_bounds(n, world, rank):
    """Contiguous split of [0, n) -- the same rule as shard() in csrc/circl_hip.hip."""
    return n * rank // world, n * (rank + 1) // world


class Ranks:
    def __init__(self, backend=None, device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        self.backend = None
        # CIRCL_DIST_FORCE_PG: create the process group even for one rank (tests/test_gpu_round4.py runs the RCCL branch that way)
        if self.world > 1 or os.environ.get("CIRCL_DIST_FORCE_PG"):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            backend = os.environ.get("CIRCL_DIST_BACKEND") or backend or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self.backend = backend

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def _tensor(self, x):
        dev = self.device if (self.device is not None and torch.cuda.is_available() and self.backend != "gloo") else "cpu"
        return torch.tensor([float(x)], dtype=torch.float64, device=dev)

    def max(self, x):
        if self.dist is None:
            return float(x)
        t = self._tensor(x)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, x):
        if self.dist is None:
            return float(x)
        t = self._tensor(x)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather(self, x):
        """per-rank values as a list indexed by rank (on every rank)"""
        if self.dist is None:
            return [float(x)]
        t = self._tensor(x)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def whole_job_rate(ranks, items_this_rank, elapsed_this_rank):
    """value = units all ranks processed / max-over-ranks time (the bench contract)."""
    total = ranks.sum(items_this_rank)
    worst = ranks.max(elapsed_this_rank)
    return total / worst, worst
