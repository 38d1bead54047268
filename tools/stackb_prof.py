"""Stack B eval forward, B=4096 bf16, 25 eager passes -- the workload of the rocprofv3 kernel-stats profile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmdeer import stackb, synth  # noqa: E402

B = int(os.environ.get("STACKB_B", "4096"))
m = stackb.CompleteDEERModel(compute_dtype=os.environ.get("STACKB_DTYPE", "bf16")).to("cuda:0").eval()
b = synth.make_batch(B, seed=1)
xs = [torch.from_numpy(b[k]).to("cuda:0") for k in ("audio", "video", "text")]
for _ in range(25):
    m(*xs)
torch.cuda.synchronize()
