#!/usr/bin/env python3
"""Eval-mode forward only (BASELINE configs[1]: B=1024 fp32; and B=4096 bf16): HIP-graph replay of model(a, v, t), samples/s."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402

dev = torch.device("cuda:0")
for B, dtype in ((1024, "fp32"), (4096, "bf16")):
    m = MultimodalDEER(ModelConfig(compute_dtype=dtype, seed=1)).to(dev).eval()
    b = synth.make_batch(B, seed=2)
    xs = [torch.from_numpy(b[k]).to(dev) for k in ("audio", "video", "text")]
    if dtype == "bf16":
        xs = [x.bfloat16() for x in xs]
    with torch.no_grad():
        for _ in range(3):
            m(*xs)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(*xs)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(*xs)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps({"workload": f"eval forward (fusion + DEER head, all outputs of forward()), B={B}, {dtype}, HIP-graph replay",
                      "ms_per_forward": round(dt * 1e3, 4), "samples_per_s": round(B / dt, 1)}))
