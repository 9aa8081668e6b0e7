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


class _Handles:
    def __init__(self, handles, after=None):
        self.handles, self.after = handles, after

    def wait(self):
        for h in self.handles:
            if h is not None:
                h.wait()
        if self.after is not None:
            self.after()
        return True


class ShardedGradSync:
    """The step's one exchange as reduce-scatter -> Adam on the owned 1/world of the parameters -> all-gather
    (SURVEY 8e / section 5): the same wire bytes as the all-reduce, but the fused Adam pass (7 streams x 57.6 MB = 403 MB
    of HBM traffic per step on every rank) shrinks to 1/world of it, and only the reduce-scatter half of the exchange sits
    between the backward and the optimizer.

    The flat buffer is cut into two REGIONS that become ready at different times -- the embedding table (95 % of the
    bytes, complete before the deferred weight-gradient GEMMs) and the rest -- and every region into `world` contiguous
    shards of `per` floats (per a multiple of 4: the Adam kernel's 16-byte accesses).  Rank r owns shard r of each region.
    A region whose length is not world * per goes through a zero-padded staging buffer (the 2.6 MB weight region; the
    table region of the bench shape divides evenly for 2 / 4 / 8 ranks and is exchanged in place).

    compress="bf16": the table region's gradient travels as bf16 (half the bytes on the wire; the sum over ranks is
    then a bf16 sum: ~2^-9 relative per element -- an opt-in whose parity delta bench.py reports, never a default).

    Usage (Model.train_step):  h = sync.start(g, 0) ... h2 = sync.start(g, 1); h.wait(); h2.wait();
                               for lo, hi, gbuf in sync.owned(): adam(flat[lo:hi], gbuf, m[lo:hi], v[lo:hi])
                               sync.gather(flat)
    """

    def __init__(self, numel: int, n_table: int, group=None, compress=None, force=False):
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.group, self.compress = group, compress
        on = dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.active = self.world > 1 or (force and on)
        self.regions = []
        for lo, hi in ((0, int(n_table)), (int(n_table), int(numel))):
            n = hi - lo
            per = -(-n // self.world)
            per = (per + 3) // 4 * 4
            a = min(lo + self.rank * per, hi)
            b = min(a + per, hi)
            self.regions.append(dict(lo=lo, hi=hi, n=n, per=per, own=(a, b), exact=(per * self.world == n), out=None, stage=None))
        assert n_table % 4 == 0, "the table region must end on a 16-byte boundary (d_model % 4 == 0)"

    # ---- helpers -------------------------------------------------------------------------------------------------
    def _via_host(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"       # test rigs only: ranks sharing one GPU

    def _buffers(self, reg, like, wire_dtype):
        per, world = reg["per"], self.world
        if reg["out"] is None or reg["out"].device != like.device or reg["out"].dtype != torch.float32:
            reg["out"] = torch.empty(per, dtype=torch.float32, device=like.device)
        need_stage = (not reg["exact"]) or wire_dtype != torch.float32
        if need_stage and (reg["stage"] is None or reg["stage"].device != like.device or reg["stage"].dtype != wire_dtype):
            reg["stage"] = torch.zeros(per * world, dtype=wire_dtype, device=like.device)
            reg["wire_out"] = torch.empty(per, dtype=wire_dtype, device=like.device)
        return need_stage

    def start(self, gflat: torch.Tensor, region: int):
        """Begin the reduce-scatter(sum) of region `region` of the flat gradient; the summed shard this rank owns lands
        in an internal fp32 buffer (see owned()).  Returns a handle with .wait() (stream-ordered under RCCL)."""
        reg = self.regions[region]
        wire = torch.bfloat16 if (self.compress == "bf16" and region == 0) else torch.float32
        staged = self._buffers(reg, gflat, wire)
        src = gflat[reg["lo"]:reg["hi"]]
        if not self.active:
            a, b = reg["own"]
            reg["out"][:b - a].copy_(gflat[a:b])
            return _Done()
        if staged:
            reg["stage"][:reg["n"]].copy_(src)                    # (casts when the wire format is bf16; the tail stays zero)
            inp, out = reg["stage"], (reg["wire_out"] if wire != torch.float32 else reg["out"])
        else:
            inp, out = src, reg["out"]
        after = None
        if wire != torch.float32:
            after = lambda: reg["out"].copy_(reg["wire_out"])
        if self._via_host(gflat):
            hin, hout = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
            dist.reduce_scatter_tensor(hout, hin, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(hout)
            if after:
                after()
            return _Done()
        h = dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return _Handles([h], after)

    def owned(self):
        """[(lo, hi, grad)]: the ranges of the flat buffer this rank updates and the summed gradient of each."""
        return [(r["own"][0], r["own"][1], r["out"][:r["own"][1] - r["own"][0]]) for r in self.regions if r["own"][1] > r["own"][0]]

    def gather(self, flat: torch.Tensor):
        """All-gather of the updated parameters: every rank's owned shards into everybody's flat buffer."""
        if not self.active:
            return flat
        for reg in self.regions:
            a, b = reg["own"]
            dst = flat[reg["lo"]:reg["hi"]]
            if reg["exact"] and not self._via_host(flat):
                dist.all_gather_into_tensor(dst, flat[a:b], group=self.group)          # in place: shard r sits at offset r * per
                continue
            per = reg["per"]
            mine = torch.zeros(per, dtype=torch.float32, device="cpu" if self._via_host(flat) else flat.device)
            mine[:b - a].copy_(flat[a:b])
            full = torch.empty(per * self.world, dtype=torch.float32, device=mine.device)
            dist.all_gather_into_tensor(full, mine, group=self.group)
            dst.copy_(full[:reg["n"]])
        return flat


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
