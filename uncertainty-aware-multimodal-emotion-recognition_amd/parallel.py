"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI via torch.distributed.

The path shards by samples (weak scaling); the only exchange is the gradient all-reduce.  The flat
gradient buffer ``mmdeer_backward`` fills is split into 3 buckets in reverse execution order
(head -> output/trimodal -> audio-visual); the library records a HIP event after each bucket is complete and the
bucket's all-reduce is enqueued on a side stream behind that event, so it overlaps the rest of backward.
Semantics are DDP's: the result is the mean over ranks of per-shard gradients (ECE / cross-dim terms are
non-linear in batch statistics, so this is not the gradient of the global-batch loss).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib


class BucketedAllReduce:
    def __init__(self, group: Optional["dist.ProcessGroup"] = None, device: Optional[torch.device] = None,
                 force: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()    # run the collectives even on a 1-rank group (rehearsal)
        lib = _lib.load()
        self.ranges = [(lib.mmdeer_bucket_begin(i), lib.mmdeer_bucket_end(i)) for i in range(3)]
        self.cuda = device is not None and device.type == "cuda"
        if self.cuda:
            self.events = [torch.cuda.Event() for _ in range(3)]
            for e in self.events:
                e.record()          # materialise the underlying hipEvent_t handles
            self.stream = torch.cuda.Stream(device=device)
        else:
            self.events, self.stream = None, None
        self._work: List = []

    def launch(self, flat: torch.Tensor) -> None:
        """Enqueue the bucket all-reduces (call right after model.train_step(..., events=self.events))."""
        if self.world == 1 and not self.force:
            return
        self._work = []
        if self.cuda:
            for (b, e), ev in zip(self.ranges, self.events):
                self.stream.wait_event(ev)
                with torch.cuda.stream(self.stream):
                    self._work.append(dist.all_reduce(flat[b:e], op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:  # gloo (CPU tests): no AVG op
            for b, e in self.ranges:
                self._work.append(dist.all_reduce(flat[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self, flat: Optional[torch.Tensor] = None) -> None:
        """Make the reduced gradients visible to the current stream (or the host for gloo)."""
        if self.world == 1 and not self.force:
            return
        for w in self._work:
            w.wait()
        if not self.cuda and flat is not None:
            flat.div_(self.world)
        self._work = []


def shard_rows(global_batch: int, rank: int, world: int):
    """Rows [begin, end) of the global batch owned by `rank` (contiguous, near-equal shards)."""
    base, rem = divmod(global_batch, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
