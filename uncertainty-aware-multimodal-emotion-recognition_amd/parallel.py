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


class RcclCommunicator:
    """An RCCL communicator behind the C ABI (``mmdeer_comm_*`` / ``mmdeer_allreduce``, include/mmdeer.h): the exchange
    of the path without torch.distributed in the data path.  Rank 0 draws the 128-byte id (``unique_id()``), the host
    distributes it, every rank constructs the communicator with its device current; ``from_torch_distributed`` does
    the distribution with a broadcast over an existing process group (bootstrap only)."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        import ctypes as C
        from . import _lib
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of mmdeer_comm_unique_id")
        self._lib = _lib.load()
        self.rank, self.world = rank, world
        handle = C.c_void_p()
        _lib.check(self._lib.mmdeer_comm_init(C.byref(handle), rank, world, unique_id))
        self._handle = handle

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().mmdeer_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, group: Optional["dist.ProcessGroup"] = None) -> "RcclCommunicator":
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0])

    def all_reduce(self, t: torch.Tensor, average: bool = True) -> None:
        """In place on the current stream (enqueue only; capturable into a HIP graph)."""
        from . import _lib
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("all_reduce: a contiguous fp32 / bf16 GPU tensor is required")
        _lib.check(self._lib.mmdeer_allreduce(t.data_ptr(), t.numel(), int(t.dtype == torch.float32), int(average),
                                              self._handle, _lib.current_stream()))

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            from . import _lib
            _lib.check(self._lib.mmdeer_comm_destroy(self._handle))
            self._handle = None


class BucketedAllReduce:
    """Gradient all-reduce of the flat buffer (the name is kept from the bucketed design; ``events`` is None: the
    library needs no per-bucket events any more, ``train_step(events=None)``).

    ``payload``: ``'bf16'`` (default on the GPU) sends the gradients as bf16 -- half the bytes over xGMI, the payload
    SURVEY 8e allows and the equivalent of torch DDP's ``bf16_compress_hook``: cast (``mmdeer_convert``), averaged
    all-reduce, cast back into the fp32 buffer; ``'fp32'`` exchanges the buffer as it is."""

    def __init__(self, group: Optional["dist.ProcessGroup"] = None, device: Optional[torch.device] = None,
                 force: bool = False, payload: Optional[str] = None, backend: Optional[str] = None):
        import os
        backend = backend or os.environ.get("MMDEER_COMM", "torch")
        if backend not in ("torch", "rccl"):
            raise ValueError("backend must be 'torch' (torch.distributed's communicator) or 'rccl' (mmdeer_comm_* of the C ABI)")
        self.backend = backend
        self._rccl: Optional[RcclCommunicator] = None
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()    # run the collective even on a 1-rank group (rehearsal)
        self.cuda = device is not None and device.type == "cuda"
        if payload is None:
            payload = "bf16" if self.cuda else "fp32"
        if payload not in ("bf16", "fp32"):
            raise ValueError("payload must be 'bf16' or 'fp32'")
        if payload == "bf16" and not self.cuda:
            raise ValueError("the bf16 payload needs the GPU (RCCL) path")
        self.payload = payload
        if backend == "rccl" and self.cuda and (self.world > 1 or self.force):
            self._rccl = RcclCommunicator.from_torch_distributed(group)
        # exact-global loss (optional): ranks exchange the loss statistics between forward and backward
        # (sum_small), each rank's gradient then is its SHARE of the global-batch gradient and the exchange is a SUM
        self.exact_global = False
        self.events = None
        self._work: List = []
        self._half: Optional[torch.Tensor] = None
        self._side: Optional["torch.cuda.Stream"] = None
        self._rccl_side: Optional[RcclCommunicator] = None

    def _convert(self, src: torch.Tensor, dst: torch.Tensor) -> None:
        from . import _lib
        lib = _lib.load()
        _lib.check(lib.mmdeer_convert(src.data_ptr(), int(src.dtype == torch.float32), dst.data_ptr(),
                                      int(dst.dtype == torch.float32), src.numel(), _lib.current_stream()))

    def launch(self, flat: torch.Tensor) -> None:
        """Enqueue the all-reduce behind the backward pass (call right after ``model.train_step``)."""
        if self.world == 1 and not self.force:
            return
        if self.cuda:   # RCCL: averaged in the collective, ordered after the backward kernels on the current stream
            if self.payload == "bf16":
                if self._half is None or self._half.numel() != flat.numel() or self._half.device != flat.device:
                    self._half = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
                self._convert(flat, self._half)
                self._work = [self._reduce(self._half)]
            else:
                self._work = [self._reduce(flat)]
        else:           # gloo (CPU tests): no AVG op
            self._work = [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        self._flat = flat

    def _reduce(self, t: torch.Tensor):
        """All-reduce of a GPU tensor (mean over ranks; SUM in exact-global mode) behind the work already on the current
        stream; returns a waitable or None."""
        avg = not self.exact_global
        if self._rccl is not None:
            self._rccl.all_reduce(t, average=avg)       # enqueued on the current stream: nothing to wait for
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group, async_op=True)

    def sum_small(self, t: torch.Tensor) -> None:
        """In-place SUM all-reduce of a small fp32 tensor (the loss statistics), ordered on the current stream."""
        if self.world == 1 and not self.force:
            return
        if self._rccl is not None:
            self._rccl.all_reduce(t, average=False)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    # ---- overlapped form (GPU): the exchange of a slice of the flat buffer on a side stream, forked from / joined to
    #      the current stream by events -- capturable into a HIP graph together with the step (model.train_step(comm=...))
    @property
    def active(self) -> bool:
        return self.cuda and (self.world > 1 or self.force)

    def launch_range(self, flat: torch.Tensor, lo: int, hi: int) -> None:
        """All-reduce ``flat[lo:hi]`` on the side stream, ordered after everything enqueued on the current stream.

        The side-stream exchange always goes through the C-ABI communicator (``mmdeer_allreduce``: enqueue-only on the
        stream it is given).  torch.distributed's NCCL work objects waited for on a side stream INSIDE a HIP-graph capture
        crash ``hipStreamEndCapture`` on this stack (torch 2.10 + ROCm 7.0 RCCL 2.26: SIGSEGV in ``capture_end``, found in
        round 3 with tools/dp_overlap_probe.py; both payloads), while the same plan through ``mmdeer_allreduce`` captures and
        replays.  The communicator is created at the first (eager, un-captured) call -- collectively, like every rank's
        first step."""
        if not self.active or hi <= lo:
            return
        if self._rccl_side is None:
            self._rccl_side = self._rccl or RcclCommunicator.from_torch_distributed(self.group)
        if self._side is None:
            self._side = torch.cuda.Stream(device=flat.device)
        self._side.wait_stream(torch.cuda.current_stream())
        avg = not self.exact_global
        with torch.cuda.stream(self._side):
            view = flat[lo:hi]
            if self.payload == "bf16":
                if self._half is None or self._half.numel() != flat.numel() or self._half.device != flat.device:
                    self._half = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
                half = self._half[lo:hi]
                self._convert(view, half)
                self._rccl_side.all_reduce(half, average=avg)
                self._convert(half, view)
            else:
                self._rccl_side.all_reduce(view, average=avg)

    def join(self) -> None:
        """The current stream waits for the exchanges launched with ``launch_range``."""
        if self.active and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def wait(self, flat: Optional[torch.Tensor] = None) -> None:
        """Make the reduced gradients visible to the current stream (or the host for gloo)."""
        if self.world == 1 and not self.force:
            return
        for w in self._work:
            if w is not None:
                w.wait()
        if self.cuda and self.payload == "bf16" and self._work:
            self._convert(self._half, self._flat)
        if not self.cuda and flat is not None and not self.exact_global:
            flat.div_(self.world)
        self._work = []


def shard_rows(global_batch: int, rank: int, world: int):
    """Rows [begin, end) of the global batch owned by `rank` (contiguous, near-equal shards)."""
    base, rem = divmod(global_batch, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
