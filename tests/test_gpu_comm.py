"""The RCCL communicator behind the C ABI (mmdeer_comm_* / mmdeer_allreduce) on a 1-rank group: one GPU box has one
GPU, and RCCL refuses two ranks on the same device, so the N > 1 arithmetic is covered by the 2-rank gloo tests of
tests/test_cpu_parallel.py and this file checks the binding, the in-place contract and graph capture."""
import pytest
import torch

from mmdeer.parallel import RcclCommunicator

pytestmark = pytest.mark.gpu


def test_rccl_communicator_single_rank_allreduce_and_capture():
    torch.cuda.set_device(0)
    uid = RcclCommunicator.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = RcclCommunicator(0, 1, uid)
    try:
        for dtype in (torch.float32, torch.bfloat16):
            x = torch.randn(1 << 20, device="cuda:0").to(dtype)
            ref = x.clone()
            comm.all_reduce(x, average=True)
            comm.all_reduce(x, average=False)
            torch.cuda.synchronize()
            assert torch.equal(x, ref)                      # one rank: sum == mean == identity
        # capturable: the collective is a plain enqueue on the capture stream
        y = torch.ones(4096, device="cuda:0")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y.mul_(2.0)
            comm.all_reduce(y)
        y.fill_(1.0)
        g.replay(); g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, torch.full_like(y, 4.0))
        with pytest.raises(ValueError):
            comm.all_reduce(torch.ones(4, device="cuda:0", dtype=torch.float64))
        comm.all_reduce(torch.empty(0, device="cuda:0"))    # empty buffer: a no-op
    finally:
        comm.close()
    with pytest.raises(ValueError, match="128 bytes"):
        RcclCommunicator(0, 1, b"short")
