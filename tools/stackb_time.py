"""Stack B eval forward timing (B sweep) on cuda:0: eager and HIP-graph replay, fp32 and bf16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mmdeer import stackb, synth

def main():
    for compute in ("bf16", "fp32"):
        m = stackb.CompleteDEERModel(compute_dtype=compute).to("cuda:0").eval()
        for B in (1, 32, 1024, 4096):
            b = synth.make_batch(B, seed=1)
            xs = [torch.from_numpy(b[k]).to("cuda:0") for k in ("audio", "video", "text")]
            for _ in range(5):
                m(*xs)
            torch.cuda.synchronize()
            n = 50
            t0 = time.perf_counter()
            for _ in range(n):
                m(*xs)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / n
            g = m.capture(*xs).graph
            torch.cuda.synchronize()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            gr = (time.perf_counter() - t0) / n
            print(f"{compute} B={B}: eager {eager*1e3:.3f} ms ({B/eager:,.0f}/s)  graph {gr*1e3:.3f} ms ({B/gr:,.0f}/s)", flush=True)

main()
