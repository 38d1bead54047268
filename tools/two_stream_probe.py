"""Probe: does the chip overlap two independent half-batch train steps?  Two model replicas with B/2 rows each run on two
streams inside ONE captured graph; compared with one replica at B and with the two half steps back to back."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402

dev = torch.device("cuda:0")


def batch(B, seed):
    d = synth.make_batch(B, seed=seed)
    a, v, t, y = (torch.from_numpy(d[k]).to(dev) for k in ("audio", "video", "text", "targets"))
    return a.bfloat16(), v.bfloat16(), t.bfloat16(), y


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def capture(models, batches, parallel):
    for m, b in zip(models, batches):
        m.train_step(*b)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    streams = [torch.cuda.Stream() for _ in models]
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if parallel:
            for s, m, b in zip(streams, models, batches):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    m.train_step(*b)
            for s in streams:
                cur.wait_stream(s)
        else:
            for m, b in zip(models, batches):
                m.train_step(*b)
    return g


B = int(os.environ.get("PROBE_B", "4096"))
full = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=1)).to(dev).train()
g = capture([full], [batch(B, 1)], False)
print(f"one step, B={B}: {timeit(g.replay):.4f} ms", flush=True)
for parts in (2, 4):
    ms = [MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=1)).to(dev).train() for _ in range(parts)]
    bs = [batch(B // parts, 10 + i) for i in range(parts)]
    gs = capture(ms, bs, False)
    print(f"{parts} steps of B={B // parts}, back to back: {timeit(gs.replay):.4f} ms", flush=True)
    gp = capture(ms, bs, True)
    print(f"{parts} steps of B={B // parts}, {parts} streams in one graph: {timeit(gp.replay):.4f} ms", flush=True)
