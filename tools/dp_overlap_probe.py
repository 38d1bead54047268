#!/usr/bin/env python3
"""1-rank probe of the overlapped data-parallel plan (two-phase backward + side-stream all-reduce) eagerly and under HIP-graph
capture, for each payload / communicator backend.  usage: dp_overlap_probe.py <payload> <backend> <mode>"""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mmdeer import synth  # noqa: E402
from mmdeer.model import ModelConfig, MultimodalDEER  # noqa: E402
from mmdeer.parallel import BucketedAllReduce  # noqa: E402

payload, backend, mode = sys.argv[1], sys.argv[2], sys.argv[3]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
B = 1024
model = MultimodalDEER(ModelConfig(compute_dtype="bf16", dropout=0.3, seed=42)).to(dev).train()
data = synth.make_batch(B, seed=42)
a, v, t, y = (torch.from_numpy(data[k]).to(dev) for k in ("audio", "video", "text", "targets"))
a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()
comm = BucketedAllReduce(device=dev, force=True, payload=payload, backend=backend)
model.train_step(a, v, t, y)
comm.launch(model.flat_grad()); comm.wait()
torch.cuda.synchronize()
print("eager single exchange ok", flush=True)
model.train_step(a, v, t, y, comm=comm)
torch.cuda.synchronize()
print("eager overlapped ok", flush=True)
if mode == "capture":
    r = model.capture_train_step(a, v, t, y, comm=comm)
    print("capture overlapped ok", flush=True)
    for _ in range(3):
        r()
    torch.cuda.synchronize()
    print("replay overlapped ok", float(r()["total_loss"]), flush=True)
dist.destroy_process_group()
