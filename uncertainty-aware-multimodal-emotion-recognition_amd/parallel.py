"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI via torch.distributed.

The path shards by samples (weak scaling); the only exchange is the gradient all-reduce.
``mmdeer_backward`` produces every weight gradient in ONE grouped launch at the end of the pass (split into
per-bucket launches each of them ran at one workgroup's latency on a mostly idle chip and cost 3 x 35 us), so all
slices of the flat gradient buffer become final together and there is nothing left to overlap a bucketed exchange
with: the exchange is a single all-reduce of the whole flat buffer (11.6 MB fp32), enqueued on the stream the
backward ran on.  Semantics are DDP's: the result is the mean over ranks of per-shard gradients (ECE / cross-dim
terms are non-linear in batch statistics, so this is not the gradient of the global-batch loss).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class BucketedAllReduce:
    """Gradient all-reduce of the flat buffer (the name is kept from the bucketed design; ``events`` is None: the
    library needs no per-bucket events any more, ``train_step(events=None)``)."""

    def __init__(self, group: Optional["dist.ProcessGroup"] = None, device: Optional[torch.device] = None,
                 force: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()    # run the collective even on a 1-rank group (rehearsal)
        self.cuda = device is not None and device.type == "cuda"
        self.events = None
        self._work: List = []

    def launch(self, flat: torch.Tensor) -> None:
        """Enqueue the all-reduce behind the backward pass (call right after ``model.train_step``)."""
        if self.world == 1 and not self.force:
            return
        if self.cuda:   # RCCL: averaged in the collective, ordered after the backward kernels on the current stream
            self._work = [dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)]
        else:           # gloo (CPU tests): no AVG op
            self._work = [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]

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
