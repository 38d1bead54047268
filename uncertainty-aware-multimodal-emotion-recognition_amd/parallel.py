"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

torch.distributed bootstraps the ranks (rendezvous, the 128-byte communicator id, barriers and the MAX of the timings); the
exchange itself goes through the C-ABI communicator (``mmdeer_comm_*``: enqueue-only RCCL calls on the stream they are
given, no torch work object -- those crash ``hipStreamEndCapture`` when waited for inside a HIP-graph capture on this stack,
DESIGN.md section 6), as ONE all-reduce of the flat gradient buffer or as reduce-scatter + all-gather (``algo='rs_ag'``: every
rank reduces 1/N of the buffer from all peers at once, then fetches the other shards -- all seven xGMI links of a GPU at the
same time, where a ring all-reduce is bound by one link).  gloo (CPU tests) runs the same arithmetic.

The path shards by samples (weak scaling); the only exchange is that of the gradients.
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

    @staticmethod
    def _checked(t: torch.Tensor, what: str) -> None:
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"{what}: a contiguous fp32 / bf16 GPU tensor is required")

    def all_reduce(self, t: torch.Tensor, average: bool = True) -> None:
        """In place on the current stream (enqueue only; capturable into a HIP graph)."""
        from . import _lib
        self._checked(t, "all_reduce")
        _lib.check(self._lib.mmdeer_allreduce(t.data_ptr(), t.numel(), int(t.dtype == torch.float32), int(average),
                                              self._handle, _lib.current_stream()))

    def reduce_scatter(self, buf: torch.Tensor, average: bool = True) -> torch.Tensor:
        """``buf`` holds world x shard elements; on return this rank's shard ``buf[rank * shard : (rank + 1) * shard]`` (returned
        as a view) is the sum / mean over ranks of that slice.  In place, enqueue only."""
        from . import _lib
        self._checked(buf, "reduce_scatter")
        if buf.numel() % self.world:
            raise ValueError("reduce_scatter: the buffer must hold world x shard elements")
        shard = buf.numel() // self.world
        mine = buf[self.rank * shard:(self.rank + 1) * shard]
        _lib.check(self._lib.mmdeer_reduce_scatter(buf.data_ptr(), mine.data_ptr(), shard, int(buf.dtype == torch.float32), int(average),
                                                   self._handle, _lib.current_stream()))
        return mine

    def all_gather(self, buf: torch.Tensor) -> None:
        """Every rank's shard ``buf[rank * shard : (rank + 1) * shard]`` is distributed to all: in place, enqueue only."""
        from . import _lib
        self._checked(buf, "all_gather")
        if buf.numel() % self.world:
            raise ValueError("all_gather: the buffer must hold world x shard elements")
        shard = buf.numel() // self.world
        mine = buf[self.rank * shard:(self.rank + 1) * shard]
        _lib.check(self._lib.mmdeer_allgather(mine.data_ptr(), buf.data_ptr(), shard, int(buf.dtype == torch.float32),
                                              self._handle, _lib.current_stream()))

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            from . import _lib
            _lib.check(self._lib.mmdeer_comm_destroy(self._handle))
            self._handle = None


def shard_elems(n: int, world: int, align: int = 8) -> int:
    """Elements per rank of an n-element buffer cut into `world` equal shards of whole 16-byte units (the tail is padding)."""
    per = -(-n // world)
    return -(-per // align) * align


class BucketedAllReduce:
    """Gradient exchange of the flat buffer (the name is kept from the bucketed design; ``events`` is None: the
    library needs no per-bucket events any more, ``train_step(events=None)``).

    ``backend``: ``'rccl'`` (default on the GPU) = the C-ABI communicator, bootstrapped over the process group;
    ``'torch'`` = torch.distributed's own collectives (always on gloo).  ``algo``: ``'allreduce'`` (default) or ``'rs_ag'``
    (reduce-scatter + all-gather over a buffer padded to world x shard elements); an attribute, so a host can time both.

    ``payload``: ``'bf16'`` (default on the GPU) sends the gradients as bf16 -- half the bytes over xGMI, the payload
    SURVEY 8e allows and the equivalent of torch DDP's ``bf16_compress_hook``: cast (``mmdeer_convert``), averaged
    all-reduce, cast back into the fp32 buffer; ``'fp32'`` exchanges the buffer as it is."""

    def __init__(self, group: Optional["dist.ProcessGroup"] = None, device: Optional[torch.device] = None,
                 force: bool = False, payload: Optional[str] = None, backend: Optional[str] = None, algo: str = "allreduce"):
        import os
        on_gpu = device is not None and device.type == "cuda"
        backend = backend or os.environ.get("MMDEER_COMM") or ("rccl" if on_gpu else "torch")
        if backend not in ("torch", "rccl"):
            raise ValueError("backend must be 'torch' (torch.distributed's communicator) or 'rccl' (mmdeer_comm_* of the C ABI)")
        if algo not in ("allreduce", "rs_ag"):
            raise ValueError("algo must be 'allreduce' or 'rs_ag' (reduce-scatter + all-gather)")
        if backend == "rccl" and not on_gpu:
            raise ValueError("the rccl backend needs a GPU device")
        self.backend = backend
        self.algo = algo
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
            # Collective set-up: if ANY rank cannot create the C-ABI communicator (librccl not loadable, ncclCommInitRank failing),
            # every rank falls back to torch.distributed's own collectives together -- the first real multi-GPU run must not die here.
            err = None
            try:
                self._rccl = RcclCommunicator.from_torch_distributed(group)
            except Exception as e:      # noqa: BLE001
                err = e
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok) == 0:
                if self._rccl is not None:
                    self._rccl.close()
                    self._rccl = None
                import sys
                print(f"[mmdeer] C-ABI communicator unavailable on some rank ({type(err).__name__ if err else 'peer'}: {err}); "
                      "gradient exchange through torch.distributed", file=sys.stderr)
                self.backend = "torch"
        # exact-global loss (optional): ranks exchange the loss statistics between forward and backward
        # (sum_small), each rank's gradient then is its SHARE of the global-batch gradient and the exchange is a SUM
        self.exact_global = False
        self.events = None
        self._work: List = []
        self._half: Optional[torch.Tensor] = None
        self._stage: Optional[torch.Tensor] = None
        self._side: Optional["torch.cuda.Stream"] = None
        self._rccl_side: Optional[RcclCommunicator] = None

    def _convert(self, src: torch.Tensor, dst: torch.Tensor) -> None:
        from . import _lib
        lib = _lib.load()
        _lib.check(lib.mmdeer_convert(src.data_ptr(), int(src.dtype == torch.float32), dst.data_ptr(),
                                      int(dst.dtype == torch.float32), src.numel(), _lib.current_stream()))

    def _staging(self, flat: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """The exchange buffer: world x shard elements (>= flat.numel(); the padding stays zero), allocated once."""
        n = self.world * shard_elems(flat.numel(), self.world)
        name = "_half" if dtype == torch.bfloat16 else "_stage"
        buf = getattr(self, name)
        if buf is None or buf.numel() != n or buf.device != flat.device:
            buf = torch.zeros(n, dtype=dtype, device=flat.device)
            setattr(self, name, buf)
        return buf

    def launch(self, flat: torch.Tensor) -> None:
        """Enqueue the exchange behind the backward pass (call right after ``model.train_step``)."""
        if self.world == 1 and not self.force:
            return
        self._flat = flat
        self._post = None        # what wait() copies back into `flat`
        n = flat.numel()
        if self.cuda:   # RCCL: averaged in the collective, ordered after the backward kernels on the current stream
            if self.payload == "bf16":
                half = self._staging(flat, torch.bfloat16)
                self._convert(flat, half[:n])
                self._work = [self._exchange(half)]
                self._post = half
            elif self.algo == "rs_ag" and n % (8 * self.world):
                stage = self._staging(flat, torch.float32)       # fp32 shards of whole 16-byte units need the padded copy
                stage[:n].copy_(flat)
                self._work = [self._exchange(stage)]
                self._post = stage
            else:
                self._work = [self._exchange(flat)]
        elif self.algo == "rs_ag":   # gloo (CPU tests): the same shard arithmetic through reduce + all_gather (no AVG op: wait() divides)
            stage = self._staging(flat, torch.float32)
            stage[:n].copy_(flat)
            shard = stage.numel() // self.world
            rank = dist.get_rank(self.group)
            for r in range(self.world):
                dist.reduce(stage[r * shard:(r + 1) * shard], dst=dist.get_global_rank(self.group, r) if self.group is not None else r,
                            op=dist.ReduceOp.SUM, group=self.group)
            mine = stage[rank * shard:(rank + 1) * shard].clone()
            dist.all_gather_into_tensor(stage, mine, group=self.group)
            self._work = []
            self._post = stage
        else:           # gloo: no AVG op
            self._work = [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]

    def _exchange(self, t: torch.Tensor):
        """The exchange of a GPU buffer (mean over ranks; SUM in exact-global mode) behind the work already on the current
        stream, by the chosen algorithm; returns a waitable or None."""
        if self.algo == "allreduce" or t.numel() % self.world:
            return self._reduce(t)
        avg = not self.exact_global
        if self._rccl is not None:
            self._rccl.reduce_scatter(t, average=avg)
            self._rccl.all_gather(t)
            return None
        shard = t.numel() // self.world
        rank = dist.get_rank(self.group)
        mine = t[rank * shard:(rank + 1) * shard]
        dist.reduce_scatter_tensor(mine, t, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group)
        return dist.all_gather_into_tensor(t, mine, group=self.group, async_op=True)

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
                half = self._staging(flat, torch.bfloat16)[lo:hi]
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
        post, target = getattr(self, "_post", None), getattr(self, "_flat", None)
        if post is not None and target is not None:
            n = target.numel()
            if post.dtype == torch.bfloat16:
                self._convert(post[:n], target)
            else:
                target.copy_(post[:n])
            self._post = None
        if not self.cuda and flat is not None and not self.exact_global:
            flat.div_(self.world)
        self._work = []


def shard_rows(global_batch: int, rank: int, world: int):
    """Rows [begin, end) of the global batch owned by `rank` (contiguous, near-equal shards)."""
    base, rem = divmod(global_batch, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
