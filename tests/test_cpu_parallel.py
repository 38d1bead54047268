"""N > 1 path on CPU: two gloo ranks exchange the flat gradient buffer (the same BucketedAllReduce object the
GPU bench uses over RCCL), and rank-sharded synthetic batches reassemble the global batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mmdeer import _lib, synth
from mmdeer.parallel import BucketedAllReduce, shard_rows


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = _lib.load().mmdeer_flat_elems()
        flat = torch.full((n,), float(rank + 1))
        flat[rank::7] += 0.5
        expect = sum(torch.full((n,), float(r + 1)).index_add_(0, torch.arange(r, n, 7), torch.full((len(range(r, n, 7)),), 0.5))
                     for r in range(world)) / world
        comm = BucketedAllReduce()
        assert comm.world == world and comm.events is None
        comm.launch(flat)
        comm.wait(flat)
        ok = torch.allclose(flat, expect)
        # exact-global mode (optional): loss statistics are SUMMED across ranks before the backward pass, and so
        # are the gradients (each rank holds its share of the global-batch gradient)
        stats = torch.arange(106, dtype=torch.float32) * (rank + 1)
        comm.sum_small(stats)
        ok = ok and torch.equal(stats, torch.arange(106, dtype=torch.float32) * sum(range(1, world + 1)))
        comm.exact_global = True
        share = torch.full((n,), float(rank + 1))
        comm.launch(share)
        comm.wait(share)
        ok = ok and torch.equal(share, torch.full((n,), float(sum(range(1, world + 1)))))
        comm.exact_global = False
        # reduce-scatter + all-gather (every rank reduces 1/N of the buffer, then fetches the others' shards): the same
        # result as the all-reduce, mean and SUM, on a buffer whose length is not a multiple of world x 8 (the tail is padding)
        from mmdeer.parallel import shard_elems
        for m in (n, 8 * world * 5 + 12):
            assert shard_elems(m, world) % 8 == 0 and world * shard_elems(m, world) >= m
            for exact in (False, True):
                rs = BucketedAllReduce(algo="rs_ag")
                rs.exact_global = exact
                g = torch.arange(m, dtype=torch.float32) * (rank + 1) + rank
                rs.launch(g)
                rs.wait(g)
                tot = sum(torch.arange(m, dtype=torch.float32) * (r + 1) + r for r in range(world))
                ok = ok and torch.allclose(g, tot if exact else tot / world)
        # weak-scaling data sharding: rank r draws rows [rB, (r+1)B) of one global stream
        B = 6
        mine = synth.make_batch(B, seed=42, row_offset=rank * B)["video"]
        gathered = [torch.zeros(B, 256) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(mine))
        whole = synth.make_batch(world * B, seed=42)["video"]
        ok = ok and np.array_equal(torch.cat(gathered).numpy(), whole)
        b, e = shard_rows(world * B, rank, world)
        ok = ok and (b, e) == (rank * B, (rank + 1) * B)
        # data-parallel training through DEERTrainer must not draw the same dropout masks on every rank:
        # identical parameters (seed), a dropout stream of its own per rank unless the caller chose one
        import tempfile

        from mmdeer.model import ModelConfig, MultimodalDEER
        from mmdeer.trainer import DEERTrainer, TrainingConfig
        tmp = tempfile.mkdtemp()
        dirs = dict(output_dir=tmp + "/o", log_dir=tmp + "/l", checkpoint_dir=tmp + "/c")
        m = MultimodalDEER(ModelConfig(seed=42))
        ok = ok and m.config.dropout_seed is None
        DEERTrainer(m, TrainingConfig(fused_optimizer=False, **dirs), device="cpu", comm=comm)
        seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(seeds, torch.tensor([m.dropout_seed], dtype=torch.int64))
        ok = ok and [int(x) for x in seeds] == [42 + 1000003 * r for r in range(world)]
        m2 = MultimodalDEER(ModelConfig(seed=42, dropout_seed=7))           # an explicit choice is kept
        DEERTrainer(m2, TrainingConfig(fused_optimizer=False, **dirs), device="cpu", comm=comm)
        ok = ok and m2.dropout_seed == 7
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_exchange(world=2):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_three_rank_gloo_gradient_exchange():
    """An odd rank count: the reduce-scatter shards do not divide the buffer."""
    test_two_rank_gloo_gradient_exchange(world=3)
