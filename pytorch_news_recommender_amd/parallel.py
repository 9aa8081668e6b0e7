"""Data parallelism for the NRMS step: one process per GPU, users sharded by rank, ONE
all-reduce(sum) of the flat fp32 gradient buffer per step (RCCL over xGMI when the backend is
"nccl"; gloo on CPU for tests).

The reference has no reachable multi-GPU path (its only call site, model/__init__.py:35-36
``P.data_parallel``, is dead code); this is new design, not a translation.  Users (batch rows)
are independent in forward and backward, the only cross-user coupling is the mean in the
cross-entropy (train_eval.py:117), so each rank scales its local gradient by 1/B_global and the
summed result is exactly the single-process gradient of the global batch.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, local_rank, world_size)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_rows(n_rows: int, rank: int, world: int):
    """Rows [lo, hi) of a global batch owned by `rank` (contiguous, near-equal)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: dict, rank: int, world: int):
    n = len(batch["browsed_titles"])
    lo, hi = shard_rows(n, rank, world)
    return {k: v[lo:hi] for k, v in batch.items()}


class _Done:
    def wait(self):
        return True


class GradAllReduce:
    """callable(flat_grad): in-place sum over ranks of the single flat gradient buffer
    (57.6 MB at V=45 800).  One large message, as the xGMI mesh prefers."""

    def __init__(self, group=None, force=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force: issue the collectives even in a one-rank group (tests/test_hip_dp.py runs the RCCL calls, their stream
        # ordering and the overlap with the deferred d(W_qkv) GEMMs on a single GPU that way)
        self.active = self.world > 1 or (force and dist.is_initialized())

    def start(self, part: torch.Tensor):
        """Begin the in-place sum of `part` over the ranks and return a handle with .wait().  With RCCL the
        collective runs on the process group's own stream, ordered after the work already enqueued on the
        current stream, and .wait() makes the current stream wait for it -- no host synchronisation: kernels
        enqueued between start() and wait() overlap with it."""
        if not self.active:
            return _Done()
        if part.is_cuda and dist.get_backend(self.group) == "gloo":
            self(part)                           # test rigs only: synchronous, staged through the host
            return _Done()
        return dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def __call__(self, flat_grad: torch.Tensor):
        if self.active:
            if flat_grad.is_cuda and dist.get_backend(self.group) == "gloo":
                # test rigs only (several ranks sharing one GPU): stage through the host
                host = flat_grad.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                flat_grad.copy_(host)
            else:
                dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return flat_grad


def broadcast_parameters(flat: torch.Tensor, src: int = 0):
    """Replicas start identical (rank `src`'s values)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if flat.is_cuda and dist.get_backend() == "gloo":
            host = flat.cpu()
            dist.broadcast(host, src=src)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src=src)
    return flat


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
