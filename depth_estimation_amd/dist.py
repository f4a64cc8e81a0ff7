"""What a data-parallel trainer needs around the hot path: the path itself shards over the batch
axis with no exchange, so this is only (1) which targets a rank owns, (2) one all-reduce for the handful of loss /
metric scalars, (3) flat-bucket all-reduce of parameter gradients (config 5).  Backend-agnostic: "nccl" (= RCCL
over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.  Nothing here touches the HIP library.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(num_targets: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the batch axis owned by `rank`.  A target travels with all of its S sources,
    depths and poses (the min over sources, metrics.py:72, is the only cross-item step and stays on one GPU).
    The first num_targets % world ranks own one target more."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if num_targets < 0:
        raise ValueError("num_targets must be >= 0")
    q, r = divmod(num_targets, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(tensors: Sequence[torch.Tensor], rank: int, world: int) -> List[torch.Tensor]:
    """Slice every tensor's leading (target) axis to this rank's shard_range."""
    n = tensors[0].shape[0]
    for t in tensors:
        if t.shape[0] != n:
            raise ValueError("all tensors must share the leading (target) axis")
    lo, hi = shard_range(n, rank, world)
    return [t[lo:hi] for t in tensors]


def reduce_scalars(values: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """SUM all-reduce of a small 1-D tensor of partial sums (loss of train.py:88 summed over the local targets,
    the logged mean of :92, the four metric sums of test.py:78-81, counts): ONE latency-bound collective per step.
    Accumulated in float64 so the result does not depend on the reduction tree beyond 1 ulp of fp32."""
    if values.dim() != 1:
        raise ValueError("reduce_scalars expects a 1-D tensor")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return values.clone()
    buf = values.detach().to(torch.float64)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf.to(values.dtype)


class GradBuckets:
    """Flat fp32 buckets over a fixed parameter list; `all_reduce()` averages the gradients across ranks.

    Sizing note for MI355X: 8 GPUs are fully connected by xGMI (7 links x ~153 GB/s each); RCCL's direct
    reduce-scatter + all-gather moves 1/8 of a bucket per peer link, so ~25 MB buckets (3 MB per link per phase)
    are past the latency knee and still leave 6 buckets of the reference's 150 MB of trainable gradients
    (pose 11.2 M + decoder 25.9 M + convgru 0.5 M fp32) to overlap with backward.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 25 << 20, average: bool = True):
        """average=True divides the summed gradients by the world size (every rank holds a replica of the same loss
        mean); average=False leaves the SUM, which is what a batch that train.py:88 SUMS over its samples needs when the
        samples are sharded over ranks."""
        self.average = average
        self._hooks, self._pending, self._handles, self._group = [], [], [], None
        self.params = [p for p in params if p.requires_grad]
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes = [], 0
        for p in self.params:
            nb = p.numel() * 4
            if cur and cur_bytes + nb > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self._flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device) for b in self.buckets]

    def all_reduce(self, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
        """Pack -> all-reduce(SUM) -> scale by 1/world -> unpack into p.grad.  With async_op the per-bucket work
        handles are returned after launch (call .wait() then `unpack()`), so buckets overlap with each other and
        with whatever the caller runs meanwhile."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        handles = []
        for flat, bucket in zip(self._flat, self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            if world > 1:
                handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op))
        if async_op:
            return handles
        self.unpack(world)
        return None


    # ---- overlapped form: all-reduce every bucket as soon as backward has produced its last gradient ----------------
    def attach(self, group: Optional[dist.ProcessGroup] = None) -> None:
        """Register one post-accumulate hook per parameter.  During backward, a bucket whose last gradient has just been
        accumulated is packed and its all-reduce launched asynchronously (RCCL runs it on its own stream), so the
        collectives of early buckets overlap with the rest of backward.  Call begin() before every backward and
        finish() after it.

        Every rank must issue the collectives in the same order, whatever order its gradients arrive in (and whether a
        parameter receives one at all: a branch not taken on one rank, an unused network): the launch order is FIXED,
        last bucket first -- backward reaches the parameters roughly in reverse order of their use, so the last bucket
        fills first -- and bucket i is launched only once every bucket after it has been.  finish() drains what is left
        in that same order."""
        if self._hooks:
            return
        self._group = group
        index = {}
        for bi, bucket in enumerate(self.buckets):
            for p in bucket:
                index[p] = bi
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(lambda q, bi=index[p]: self._grad_ready(bi)))

    def detach(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def begin(self) -> None:
        self._pending = [len(b) for b in self.buckets]
        self._handles = [None] * len(self.buckets)
        self._cursor = len(self.buckets) - 1          # next bucket to launch (descending)
        self._active = True

    def _launch(self, bi: int) -> None:
        flat, off = self._flat[bi], 0
        for p in self.buckets[bi]:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        if dist.is_initialized() and dist.get_world_size(self._group) > 1:
            self._handles[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self._group, async_op=True)
        else:
            self._handles[bi] = False

    def _grad_ready(self, bi: int) -> None:
        if not getattr(self, "_active", False):
            return      # a backward outside begin()/finish(): gradients stay local (finish() without begin() raises)
        self._pending[bi] -= 1
        while self._cursor >= 0 and self._pending[self._cursor] <= 0:
            self._launch(self._cursor)
            self._cursor -= 1

    def finish(self) -> None:
        """Launch the buckets backward never completed (parameters without a gradient this step) in the fixed order,
        wait for every collective and write the reduced gradients back into p.grad."""
        if not getattr(self, "_active", False):
            raise RuntimeError("GradBuckets.finish() without begin(): call begin() before backward so that the hooks count "
                               "this step's gradients")
        while self._cursor >= 0:
            self._launch(self._cursor)
            self._cursor -= 1
        self._active = False
        for h in self._handles:
            if h:
                h.wait()
        self.unpack()

    def nbytes(self) -> int:
        return sum(f.numel() for f in self._flat) * 4

    def unpack(self, world: Optional[int] = None):
        if world is None:
            world = dist.get_world_size(self._group) if dist.is_initialized() else 1
        if not self.average:
            world = 1
        for flat, bucket in zip(self._flat, self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                g = flat[off:off + n].reshape(p.shape) / world
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
