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
        # reduce-scatter / all-gather (in place; one rank: identities), also inside a capture
        z = torch.arange(1 << 12, device="cuda:0", dtype=torch.float32)
        mine = comm.reduce_scatter(z, average=True)
        comm.all_gather(z)
        torch.cuda.synchronize()
        assert mine.data_ptr() == z.data_ptr() and torch.equal(z, torch.arange(1 << 12, device="cuda:0", dtype=torch.float32))
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            z.add_(1.0)
            comm.reduce_scatter(z, average=False)
            comm.all_gather(z)
        g2.replay(); g2.replay()
        torch.cuda.synchronize()
        assert torch.equal(z, torch.arange(1 << 12, device="cuda:0", dtype=torch.float32) + 2.0)
        with pytest.raises(ValueError):
            comm.all_reduce(torch.ones(4, device="cuda:0", dtype=torch.float64))
        comm.all_reduce(torch.empty(0, device="cuda:0"))    # empty buffer: a no-op
    finally:
        comm.close()
    with pytest.raises(ValueError, match="128 bytes"):
        RcclCommunicator(0, 1, b"short")


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bucketed_allreduce_gpu_paths_on_a_one_rank_group():
    """parallel.BucketedAllReduce on the GPU (a forced 1-rank RCCL group): both communicator backends, both payloads,
    the mean / SUM (exact-global) modes, the side-stream range form and the statistics exchange leave a 1-rank buffer
    unchanged (up to the bf16 round trip of the payload)."""
    import torch.distributed as dist

    from mmdeer import _lib
    from mmdeer.parallel import BucketedAllReduce

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n = int(_lib.load().mmdeer_flat_elems())
        ref = torch.randn(n, device="cuda:0")
        assert BucketedAllReduce(device=torch.device("cuda", 0), force=True).backend == "rccl"     # the GPU default: the C-ABI communicator
        for backend in ("torch", "rccl"):
            for payload in ("bf16", "fp32"):
                comm = BucketedAllReduce(device=torch.device("cuda", 0), force=True, payload=payload, backend=backend)
                assert comm.active and comm.world == 1 and comm.backend == backend
                for exact, algo in ((False, "allreduce"), (True, "allreduce"), (False, "rs_ag"), (True, "rs_ag")):
                    comm.exact_global = exact
                    comm.algo = algo
                    flat = ref.clone()
                    comm.launch(flat)
                    comm.wait(flat)
                    lo = n // 3 // 4 * 4
                    comm.launch_range(flat, lo, n)
                    comm.launch_range(flat, 0, lo)
                    comm.join()
                    torch.cuda.synchronize()
                    if payload == "fp32":
                        assert torch.equal(flat, ref)
                    else:
                        assert torch.allclose(flat, ref, rtol=2 ** -7, atol=0)        # two bf16 round trips
                stats = torch.arange(106, dtype=torch.float32, device="cuda:0")
                comm.sum_small(stats)
                assert torch.equal(stats.cpu(), torch.arange(106, dtype=torch.float32))
                if comm._rccl is not None:
                    comm._rccl.close()
        with pytest.raises(ValueError):
            BucketedAllReduce(device=torch.device("cuda", 0), backend="mpi")
    finally:
        dist.destroy_process_group()


def test_step_capture_with_a_live_process_group_and_an_in_graph_exchange():
    """ADVICE r3 / DESIGN section 6: a HIP-graph capture of the training step while a process group is alive (its watchdog thread polls
    events of earlier eager collectives; in the default 'global' capture mode such a call from another thread invalidates a capture
    that is in progress -- the one unexplained exit-1 rehearsal of round 3).  capture_train_step captures in 'thread_local' mode then;
    here: eager collectives first (work for the watchdog), then captures with both exchange algorithms inside, and replays."""
    import torch.distributed as dist

    from mmdeer import synth
    from mmdeer.model import ModelConfig, MultimodalDEER
    from mmdeer.parallel import BucketedAllReduce

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        dev = torch.device("cuda", 0)
        m = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=3)).to(dev).train()
        b = synth.make_batch(300, seed=5)
        a, v, t, y = (torch.from_numpy(b[k]).to(dev) for k in ("audio", "video", "text", "targets"))
        comm = BucketedAllReduce(device=dev, force=True)
        for _ in range(4):                               # eager collectives of torch.distributed itself: the watchdog has work
            dist.all_reduce(torch.ones(1024, device=dev))
        m.train_step(a, v, t, y)

        def exchange():
            comm.launch(m.flat_grad())
            comm.wait()
        for algo in ("allreduce", "rs_ag"):
            comm.algo = algo
            exchange()                                   # staging buffers are allocated outside the capture
            replay = m.capture_train_step(a, v, t, y, after=exchange)
            l0 = float(replay()["total_loss"])
            dist.all_reduce(torch.ones(8, device=dev))   # ... and between replays
            l1 = float(replay()["total_loss"])
            torch.cuda.synchronize()
            assert l0 == l0 and l1 == l1 and abs(l0 - l1) < 1.0     # fresh dropout masks per replay, same batch
            g = m.flat_grad()
            assert torch.isfinite(g).all() and float(g.abs().sum()) > 0
    finally:
        dist.destroy_process_group()


def test_stack_b_trainer_with_a_gradient_exchange(tmp_path):
    """Data parallel for Stack B's fused training step: DEERTrainer exchanges the model's flat gradient buffer between the (replayed)
    step and FlatAdamW.  On a 1-rank group the mean over ranks is the identity, so the trainer with the communicator (both exchange
    algorithms, eager and graph mode) must end with exactly the parameters of the trainer without one."""
    import copy

    import torch.distributed as dist
    from torch.utils.data import DataLoader, TensorDataset

    from mmdeer import stackb, synth
    from mmdeer.parallel import BucketedAllReduce
    from mmdeer.trainer import DEERTrainer, TrainingConfig

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        dev = torch.device("cuda", 0)
        b = synth.make_batch(96, seed=8)
        ds = TensorDataset(*(torch.from_numpy(b[k]) for k in ("audio", "video", "text", "targets")))
        m0 = stackb.CompleteDEERModel(stackb.ModelConfig(), compute_dtype="bf16").to(dev)
        runs = []
        for tag, algo, graph in (("none", None, False), ("ar", "allreduce", False), ("rs", "rs_ag", True)):
            m = copy.deepcopy(m0)
            comm = None
            if algo is not None:
                comm = BucketedAllReduce(device=dev, force=True, payload="fp32")
                comm.algo = algo
            cfg = TrainingConfig(batch_size=32, num_epochs=1, output_dir=str(tmp_path / f"o{tag}"), log_dir=str(tmp_path / f"l{tag}"),
                                 checkpoint_dir=str(tmp_path / f"c{tag}"), use_graph=graph)
            tr = DEERTrainer(m, cfg, device=dev, comm=comm)
            assert tr.fused_b
            loaders = {"iemocap": DataLoader(ds, batch_size=32, shuffle=False)}
            losses = [tr.train_epoch(loaders)["total_loss"] for _ in range(2)]
            runs.append((losses, m))
        for losses, m in runs[1:]:
            assert losses == pytest.approx(runs[0][0], rel=1e-6)
            for (n, p1), (_, p2) in zip(runs[0][1].named_parameters(), m.named_parameters()):
                assert torch.equal(p1, p2), n
        with pytest.raises(NotImplementedError):
            m = copy.deepcopy(m0)
            DEERTrainer(m, TrainingConfig(fused_optimizer=False, output_dir=str(tmp_path / "x"), log_dir=str(tmp_path / "y"), checkpoint_dir=str(tmp_path / "z")),
                        device=dev, comm=BucketedAllReduce(device=dev, force=True))      # the autograd route has no flat gradient buffer
    finally:
        dist.destroy_process_group()
