#!/usr/bin/env python3
"""bench.py -- samples/sec of one training step (forward + MultiTaskDEERLoss + backward, gradients
materialised) of the fusion + DEER path at B = 4096 per GPU, bf16 MFMA / fp32 accumulate
(BASELINE.json configs[2]; with --gpus N the same per-GPU batch, data parallel: configs[3]).

Contract: W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both
sides, MAX over ranks, rank 0 prints ONE JSON line.  `roofline` is the trimodal in_proj GEMM
(M = 2B, K = 512, N = 1536), timed in situ with HIP events recorded around its launch on the launch
stream.  `cpu_baseline` is the CPU oracle's train step (a port, validated against the imported
reference by tests/test_oracle_golden.py) on the host cores, N = 1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # see mmdeer/_lib.py
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_MFMA_PEAK = 2.5e15   # dense bf16 FLOP/s, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
F32_MFMA_PEAK = 157.3e12


def kernel_sources_sha():
    """sha256 over the kernel sources the library is built from: ties a PMC record to the code it was measured on."""
    import hashlib
    d = os.path.join(ROOT, "uncertainty-aware-multimodal-emotion-recognition_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


PROFILE_TAG = "r04"      # profiles/<tag>_* : the committed records of this round (tools/collect_profiles.py writes them)


def _file_stamp(path):
    import hashlib
    st = os.stat(path)
    return {"file": os.path.relpath(path, ROOT), "mtime": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(st.st_mtime)),
            "sha256": hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]}


def measured_traffic(dtype, batch):
    """HBM-side bytes per launch of the roofline kernel from the committed PMC record (separate rocprofv3 --pmc passes,
    FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes; tools/pmc_fused_summary.py writes the record).  None
    unless the record was taken on exactly these kernel sources, dtype and batch."""
    try:
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        rec = json.load(open(path))
        if rec.get("src_sha") == kernel_sources_sha() and rec.get("dtype") == dtype and rec.get("batch") == batch:
            return float(rec["traffic_bytes"]), _file_stamp(path)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def recorded_profile(batch, dtype):
    """Figures READ FROM COMMITTED FILES, not measured by this run: the rocprofv3 --kernel-trace --stats summary
    of this command (profiles/<tag>_bench_kernel_stats.csv + .json naming the sources / batch it was taken on) -> average
    duration of the roofline kernel and GB/s of the HBM-bound row kernels against their algorithmic bytes; and the MFMA
    calibration probe of the box the profiles were taken on.  Each part is present only while its record matches the
    kernel sources of this build, and carries the file's name, mtime and hash."""
    import csv
    import re
    out = {}
    try:
        mpath = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_bench_kernel_stats.json")
        meta = json.load(open(mpath))
        if meta.get("src_sha") == kernel_sources_sha() and meta.get("batch") == batch and meta.get("dtype") == dtype:
            cpath = os.path.join(ROOT, "profiles", meta["csv"])
            avg = {}
            for r in csv.DictReader(open(cpath)):
                n = r["Name"].replace("void ", "").replace("mmdeer::(anonymous namespace)::", "").split("(")[0]
                avg[n] = float(r["AverageNs"]) * 1e-9
            B, s = batch, 2     # bf16 activations
            # algorithmic bytes per launch of the HBM-bound kernels (activations in + out; statistics / parameters are noise)
            hbm = {"ln_fwd_kernel<false, 2, true>": 2 * B * 512 * s, "ln_fwd_kernel<false, 1, true>": 2 * B * 256 * s,
                   "ln_bwd_kernel<false, 2, true>": 3 * B * 512 * s, "ln_bwd_kernel<false, 1, true>": 3 * B * 256 * s,
                   "nig_fwd_kernel<false>": B * 192 * s + B * 12 * 4 + 7 * B * 3 * 4, "nig_bwd_kernel<false>": 2 * B * 192 * s + B * 12 * 4 + B * 3 * 4,
                   "nig_fused_kernel<false>": 2 * B * 192 * s + B * 12 * 4 + 7 * B * 3 * 4,
                   "tri_fused_kernel<1>": 2 * B * 512 * s + 1536 * 512 * s + B * 512 * s + B * 32 * 4 + 2 * B * 1536 * s}
            if "reduce_bytes" in meta:      # the weight-gradient fold: slabs in + gradient out (tools/collect_profiles.py computes it)
                hbm["reduce_partials_kernel"] = int(meta["reduce_bytes"])
            rec = {"source": "committed profile", **_file_stamp(cpath), "src_sha": meta["src_sha"], "hbm_gbps": {}}
            k = "tri_fused_kernel<0>"
            if k in avg:
                rec["roofline_kernel_avg_us"] = round(avg[k] * 1e6, 2)
                rec["roofline_frac"] = round(2.0 * (2 * B) * 512 * 1536 / avg[k] / BF16_MFMA_PEAK, 4)
            # the two kernels that dominate the step by time, against the MFMA peak (flops from the layer shapes)
            w_sq = 3 * 64 * 128 + 384 * 256 + 256 * 256 + 256 * 512 + 3 * 512 * 512 + 2 * 1536 * 512 + 512 * 256 + 512 * 768 + \
                256 * 512 + 2 * 2 * 256 * 256 + 256 * 256 + 256 * 128      # sum of N x K over the weight-gradient problems, 2B-row ones twice
            chain_w = {"F1-F6": 256 * 256 + 256 * 128 + 512 * 768 + 2 * 2 * 256 * 256 + 256 * 512 + 512 * 256,
                       "F9-F17": 3 * 512 * 512 + 256 * 512 + 256 * 256 + 384 * 256 + 3 * 64 * 128,
                       "B2-B10": 3 * 64 * 128 + 384 * 256 + 256 * 256 + 256 * 512 + 3 * 512 * 512,
                       "B13-B17": 512 * 256 + 256 * 512 + 2 * 2 * 256 * 256}
            nchain = len(chain_w)
            rec["mfma_tflops"] = {}
            for kname, fl in (("gemm_tt_dma_kernel<128, 128, 2>", 2.0 * B * w_sq), ("gemm_tt_dma_kernel<128, 128, 1>", 2.0 * B * w_sq),
                              ("gemm_tt_dma_kernel<256, 256, 1>", 2.0 * B * w_sq),
                              ("chain_kernel_s16", 2.0 * B * sum(chain_w.values()) / nchain),
                              ("chain_kernel_s32", 2.0 * B * sum(chain_w.values()) / nchain)):
                if kname in avg:
                    rec["mfma_tflops"][kname] = {"avg_us": round(avg[kname] * 1e6, 2), "tflops": round(fl / avg[kname] / 1e12, 1),
                                                 "frac_of_mfma_peak": round(fl / avg[kname] / BF16_MFMA_PEAK, 4),
                                                 "note": ("all weight gradients of the step in one launch" if kname.startswith("gemm_tt")
                                                          else "average over the four layer-chain launches (F1-F6 incl. the input projections, F9-F17, B2-B10, B13-B17)")}
            for name, nbytes in hbm.items():
                if name in avg:
                    rec["hbm_gbps"][name] = {"gbps": round(nbytes / avg[name] / 1e9, 1), "frac_of_8TBps": round(nbytes / avg[name] / 8e12, 3),
                                             "avg_us": round(avg[name] * 1e6, 2), "algorithmic_bytes": nbytes}
            out["rocprof"] = rec
    except (OSError, ValueError, KeyError):
        pass
    try:
        cal = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_calibration.txt")
        m = re.search(r"MFMA bf16 .*?: ([0-9.]+) TFLOP/s", open(cal).read())
        if m and dtype == "bf16":
            out["calibration"] = {"source": "committed profile (another box than this run's)", **_file_stamp(cal),
                                  "mfma_bf16_tflops_back_to_back": float(m.group(1))}
    except OSError:
        pass
    return out or None


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(batch, seconds=12.0):
    """Time the oracle's fwd + loss + bwd (torch-CPU fp32) on all host cores for ~`seconds`."""
    from mmdeer import synth
    from oracle import deer_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    b = {k: torch.from_numpy(v) for k, v in synth.make_batch(batch, seed=42).items()}
    P = O.to_params(synth.reference_init_state(include_gate=False), requires_grad=True)
    g = torch.Generator().manual_seed(0)

    def masks():
        k = lambda *s: (torch.rand(*s, generator=g) < 0.7)  # noqa: E731
        return {"av_attn_a2v": k(batch, 8), "av_attn_v2a": k(batch, 8), "av_fuse": k(batch, 256),
                "tri_attn": k(batch, 8, 2, 2), "tri_fuse": k(batch, 512), "out_proj": k(batch, 512),
                "fp0": k(batch, 256), "fp1": k(batch, 256), "ev0": k(batch, 3, 128), "ev1": k(batch, 3, 64)}

    def step():
        O.train_step(P, b["audio"], b["video"], b["text"], b["targets"], masks=masks(), p=0.3)

    for _ in range(2):
        step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= 3:
            break
    return {"value": round(batch * n / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps (fwd + MultiTaskDEERLoss + bwd, dropout masks drawn per step) at B={batch}, "
                      f"torch-CPU fp32, {dt:.1f} s"}


def stackb_cpu_baseline(batch, seconds=10.0):
    """Time the oracle's Stack B eval forward (torch-CPU fp32, no_grad) on all host cores for ~`seconds`."""
    from mmdeer import stackb, synth
    from oracle import deer_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    P = {k: v.detach().clone() for k, v in stackb.CompleteDEERModel().state_dict().items()}
    b = synth.make_batch(batch, seed=42)
    xs = [torch.from_numpy(b[k]) for k in ("audio", "video", "text")]
    with torch.no_grad():
        O.stackb_forward(P, *xs)
        n, t0 = 0, time.perf_counter()
        while True:
            O.stackb_forward(P, *xs)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds and n >= 3:
                break
    return {"value": round(batch * n / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} eval forwards of CompleteDEERModel at B={batch}, torch-CPU fp32, {dt:.1f} s"}


def bench_stackb(args, dev, world, rank):
    """--workload stackb_infer: samples/sec of the Stack B (complete_project.CompleteDEERModel) eval forward, replayed
    as a HIP graph; every rank serves its own batch (replicas, no collective)."""
    from mmdeer import stackb, synth
    B, K, W = args.batch, args.steps, args.warmup
    model = stackb.CompleteDEERModel(compute_dtype=args.dtype).to(dev).eval()
    data = synth.make_batch(B, seed=42, row_offset=rank * B)
    xs = [torch.from_numpy(data[k]).to(dev) for k in ("audio", "video", "text")]
    step = (lambda: model(*xs)) if args.eager else model.capture(*xs).graph.replay
    for _ in range(W):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    if rank != 0:
        return
    cfg = model.config
    macs = ((cfg.audio_dim + cfg.video_dim + cfg.text_dim) * 256 + 3 * (cfg.encoder_layers + 1) * 256 * 256       # encoders
            + 3 * (256 * 512 + 2 * 256 * 256 + 256 * 128 + 128 * 64) + 768 * 256                                   # attention
            + 512 * 512 * 2 + 768 * 512 * 2 + 512 * 512 + 512 * 768 + 3 * (256 * 128 + 128 * 4))                   # fusion, heads
    flops = 2.0 * macs * B
    peak = BF16_MFMA_PEAK if args.dtype == "bf16" else F32_MFMA_PEAK
    out = {"metric": "samples/sec (Stack B CompleteDEERModel eval forward)", "value": round(world * B * K / dt, 1), "unit": "samples/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"complete_project.CompleteDEERModel eval forward, B={B} per GPU, Xavier-initialised weights",
                      "launch": "eager" if args.eager else "hip-graph replay", "parallelism": f"replicas x{world}"},
           "roofline": {"bound": "mfma", "achieved": round(flops * K / dt / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                        "frac": round(flops * K / dt / peak, 4), "traffic": None,
                        "note": "whole forward (25 launches), algorithmic GEMM flops / wall time; the layers are 4-14 us launches bound by fixed launch cost, not MFMA"}}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = stackb_cpu_baseline(B)
    print(json.dumps(out))


def bench_stackb_train(args, dev, world, rank):
    """--workload stackb_train: samples/sec of the Stack B (complete_project.CompleteDEERModel) training step -- forward with dropout +
    MultiTaskDEERLoss + backward as ONE HIP graph (the sample-local layer runs as launches of the layer-chain kernel in bf16), and
    beside it the same step with the eager FlatAdamW optimiser step, the launch-by-launch plan of the same model on the same box
    (--dtype bf16 only).  With more than one rank (or the 1-rank rehearsal MMDEER_FORCE_COMM=1) every step ends with the exchange of the
    model's flat gradient buffer, host-enqueued behind the replayed graph (what DEERTrainer does for Stack B): mean over the ranks."""
    import copy

    from mmdeer import stackb, synth
    from mmdeer.optim import FlatAdamW
    B, K, W = args.batch, args.steps, args.warmup
    model = stackb.CompleteDEERModel(compute_dtype=args.dtype).to(dev).train()
    twin = copy.deepcopy(model) if args.dtype == "bf16" else None
    data = synth.make_batch(B, seed=42, row_offset=rank * B)
    a, v, t, y = (torch.from_numpy(data[k]).to(dev) for k in ("audio", "video", "text", "targets"))
    if args.dtype == "bf16":
        a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()        # bf16 feature blocks resident in HBM, as the north-star line

    force_comm = os.environ.get("MMDEER_FORCE_COMM") == "1" and "RANK" in os.environ
    comm = None
    if world > 1 or force_comm:
        from mmdeer.parallel import BucketedAllReduce
        comm = BucketedAllReduce(device=dev, force=force_comm, payload=args.grad_comm)

    def timed(m, with_opt):
        opt = FlatAdamW(m, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0)
        graph_step = m.capture_train_step_fused(a, v, t, y)
        flat_g = m._flat(dev)["g"]

        def rep():
            ld = graph_step()
            if comm is not None:
                comm.launch(flat_g)
                comm.wait(flat_g)
            return ld
        for _ in range(max(W, 3)):
            rep(); opt.step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            ld = rep()
            if with_opt:
                opt.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt), float(ld["total_loss"]), rep

    dt, loss, rep = timed(model, False)
    dt_opt, _, _ = timed(model, True)
    extra = {}
    if twin is not None:
        twin.train_plan = "ops"
        dt_ops, _, _ = timed(twin, False)
        extra["launch_by_launch_ms_per_step"] = round(dt_ops / K * 1e3, 4)
    if rank != 0:
        return
    cfg = model.config
    macs = ((cfg.audio_dim + cfg.video_dim + cfg.text_dim) * 256 + 3 * (cfg.encoder_layers + 1) * 256 * 256
            + 3 * (256 * 512 + 2 * 256 * 256 + 256 * 128 + 128 * 64) + 768 * 256
            + 512 * 512 * 2 + 768 * 512 * 2 + 512 * 512 + 512 * 768 + 3 * (256 * 128 + 128 * 4))
    flops = 3 * 2.0 * macs * B
    peak = BF16_MFMA_PEAK if args.dtype == "bf16" else F32_MFMA_PEAK
    st = getattr(model, "_flat_state", {}) or {}
    out = {"metric": "samples/sec (Stack B CompleteDEERModel training step: fwd + MultiTaskDEERLoss + bwd)", "value": round(world * B * K / dt, 1),
           "unit": "samples/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"complete_project.CompleteDEERModel (reference run_multimodal_deer.py:231-247) training step, dropout {cfg.dropout}, "
                                  f"B={B} per GPU, Xavier-initialised weights; optimiser step excluded from `value` and reported beside it",
                      "launch": "hip-graph replay", "parallelism": f"replicas x{world}",
                      "plan": "layer chains (mmdeer_chain)" if st.get("frag") is not None else "launch by launch"},
           "grad_exchange": "none" if comm is None else f"{comm.algo} of the flat gradient buffer ({comm.payload} payload), host-enqueued behind the graph replay",
           "train_step_with_optimizer_ms": round(dt_opt / K * 1e3, 4), "final_loss": round(loss, 6),
           "roofline": {"bound": "mfma", "achieved": round(flops * K / dt / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                        "frac": round(flops * K / dt / peak, 4), "traffic": None,
                        "note": "whole step, algorithmic GEMM flops (3 x forward) / wall time; a step of small layers: bound by the layer ends of its "
                                "chains and by fixed launch cost, not by MFMA (DESIGN.md section 4, Stack B training)"}}
    out.update(extra)
    print(json.dumps(out))


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` (N > 1, or the 1-rank rehearsal MMDEER_FORCE_COMM=1, not under a launcher): start N rank
    processes -- one per GPU -- as CHILD processes of this one, which has not touched the GPU, and return their exit code.  The
    same command line the driver uses for its own N > 1 runs (torch.distributed.run, 127.0.0.1 rendezvous) on a port the kernel
    hands out at this moment (the tools that rehearse the data-parallel path go through here: a fixed --master-port reused by
    back-to-back runs is one way to an exit status 1 with the reason on a stderr nobody kept)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def plumbing(args, world, rank):
    """--plumbing: the launch / rendezvous / timing protocol of the N-rank bench on the CPU (gloo), no GPU work: what
    tests/test_cpu_bench_spawn.py runs.  Same barrier + MAX-over-ranks bracket, same single JSON line from rank 0, and the
    same `data_parallel` diagnostics block as the GPU line (exchange timing, plan, per-rank step times, rank count as the
    process group reports it, payload bytes) computed on a flat gradient-sized CPU tensor."""
    from mmdeer.parallel import BucketedAllReduce
    from mmdeer.spec import param_offsets
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: world size {dist.get_world_size()} != --gpus {args.gpus}")
        dist.barrier()
    flat = torch.full((param_offsets()[1],), float(rank + 1))
    comm = BucketedAllReduce(device=torch.device("cpu"), payload="fp32") if world > 1 else None
    xs = []
    for _ in range(3):
        t = time.perf_counter()
        if comm:
            comm.launch(flat)
            comm.wait(flat)
        xs.append((time.perf_counter() - t) * 1e6)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if comm:
            comm.launch(flat)
            comm.wait(flat)
    own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    tt = torch.tensor([own], dtype=torch.float64)
    per_rank = [round(own / max(args.steps, 1) * 1e3, 4)]
    xus = torch.tensor([sorted(xs)[1]], dtype=torch.float64)
    if world > 1:
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank = [round(float(x) / max(args.steps, 1) * 1e3, 4) for x in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(xus, op=dist.ReduceOp.MAX)
        assert abs(float(flat[0]) - (world + 1) / 2) < 1e-5 * world, "mean over ranks is preserved by repeated averaging"
    if rank == 0:
        out = {"plumbing": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(float(tt) / max(args.steps, 1) * 1e3, 4)}
        if world > 1:
            out["data_parallel"] = {"nranks": dist.get_world_size(), "backend": dist.get_backend(), "payload": "fp32",
                                    "payload_bytes": flat.numel() * 4, "exchange_us": round(float(xus), 1), "plan": "host-enqueued",
                                    "why": "CPU plumbing run (gloo): no HIP graph", "per_rank_ms_per_step": per_rank}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--plumbing", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="samples per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grad-comm", default="bf16", choices=["bf16", "fp32"],
                    help="payload of the data-parallel gradient all-reduce (N > 1): bf16 halves the bytes over xGMI")
    ap.add_argument("--workload", default="train_step", choices=["train_step", "stackb_infer", "stackb_train"],
                    help="train_step: the north-star line (default); stackb_infer / stackb_train: SURVEY 8f-1, CompleteDEERModel eval forward / training step")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying the captured HIP graph")
    ap.add_argument("--no-autotune", action="store_true",
                    help="keep the library's default launch plan instead of timing the plans on this GPU first (MultimodalDEER.autotune_launch_plan)")
    args = ap.parse_args()

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    rehearsal = os.environ.get("MMDEER_FORCE_COMM") == "1" and not args.plumbing
    if (args.gpus > 1 or rehearsal) and not under_launcher:
        # N ranks as child processes, started BEFORE anything here touches the GPU (device_count does not initialise it)
        if not args.plumbing and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a "
                         f"{world}-rank number as n_gpus={args.gpus}")
    if args.plumbing:
        return plumbing(args, world, rank)
    force_comm = os.environ.get("MMDEER_FORCE_COMM") == "1" and "RANK" in os.environ   # 1-rank rehearsal of the DP path
    if world > 1 or force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        if dist.get_world_size() != world:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, expected {world}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if args.workload in ("stackb_infer", "stackb_train"):
        (bench_stackb if args.workload == "stackb_infer" else bench_stackb_train)(args, dev, world, rank)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    from mmdeer import synth
    from mmdeer.model import ModelConfig, MultimodalDEER
    from mmdeer.parallel import BucketedAllReduce

    B = args.batch
    # same parameters on every rank (seed), a dropout stream of its own per rank (dropout_seed): uncorrelated masks across shards
    model = MultimodalDEER(ModelConfig(compute_dtype=args.dtype, dropout=0.3, seed=42, dropout_seed=42 + 1000003 * rank)).to(dev).train()
    data = synth.make_batch(B, seed=42, row_offset=rank * B)       # rank r owns rows [rB, (r+1)B) of the global stream
    a, v, t, y = (torch.from_numpy(data[k]).to(dev) for k in ("audio", "video", "text", "targets"))
    if args.dtype == "bf16":
        a, v, t = a.bfloat16(), v.bfloat16(), t.bfloat16()         # BASELINE configs 3-5: bf16 feature blocks
    comm = BucketedAllReduce(device=dev, force=force_comm, payload=args.grad_comm) if (world > 1 or force_comm) else None
    # MMDEER_EXACT_GLOBAL=1 (optional): the 106 loss statistics are summed across ranks between forward and
    # backward, so every rank optimises the loss of the GLOBAL batch; default is DDP semantics (mean of per-shard losses)
    exact = comm is not None and os.environ.get("MMDEER_EXACT_GLOBAL", "0") == "1"
    if exact:
        comm.exact_global = True
    sc = dict(stats_comm=comm) if exact else {}
    K, W = args.steps, args.warmup
    prof = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for e0, e1 in prof:
        e0.record(); e1.record()

    from mmdeer import _lib
    from mmdeer.optim import FusedAdamW
    opt = FusedAdamW(model, lr=1e-4, weight_decay=1e-5, eps=1e-8, max_grad_norm=1.0)   # reference trainer settings

    def timed(fn, n):
        """ms per call of `fn` over n calls, bracketed like the metric (barrier + synchronize, MAX over ranks)."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt) / n * 1e3

    # Launch plan: the layer chains win or lose against the separate launches depending on the box (their bound, the L2 -> CU
    # path, varies far more between boxes than MFMA or HBM rates): the plans are timed on this GPU before anything else and the
    # fastest becomes the library's plan for the run (untimed set-up, like the data-parallel plan choice below).  An explicit
    # MMDEER_CHAIN / MMDEER_CHAIN_BWD in the environment (the A/B tools) or --no-autotune keeps the plan as set.
    launch_plan = None
    if not args.eager and not args.no_autotune and not any(k in os.environ for k in ("MMDEER_CHAIN", "MMDEER_CHAIN_BWD")):
        def rmax(x):
            if world == 1:
                return x
            tt = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt)
        try:
            launch_plan = model.autotune_launch_plan(a, v, t, y, reduce_max=rmax, **sc)
        except Exception as e:               # noqa: BLE001  -- the default plan is always a valid one
            print(f"[bench] launch-plan autotuning failed ({type(e).__name__}: {e}); keeping the library's default plan", file=sys.stderr)
            torch.cuda.synchronize()
            launch_plan = {"plan": "as set", "options": {k: _lib.get_option(k) for k in ("chain", "chain_bwd", "chain_nig")},
                           "why": f"autotuning failed: {type(e).__name__}"}
    # One step = ~35 kernels of 4-40 us: launched one by one the host needs about as long as the GPU, so the step is
    # captured once into a HIP graph (dropout masks advance through a device-side counter) and replayed.
    ev = comm.events if comm else None
    comm_in_graph = False
    comm_mode = "none"
    replay = None
    dp = None        # diagnostics of the data-parallel exchange (world > 1 or the 1-rank rehearsal)
    if comm:
        def exchange():
            comm.launch(model.flat_grad())
            comm.wait()

        def exchange_us(algo):
            """One eager exchange with `algo` (allocates its staging buffer outside any capture), then the median of 10 timed ones."""
            comm.algo = algo
            exchange()
            torch.cuda.synchronize()
            xe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            if world > 1:
                dist.barrier()
            for e0, e1 in xe:
                e0.record(); exchange(); e1.record()
            torch.cuda.synchronize()
            x = torch.tensor([sorted(e0.elapsed_time(e1) for e0, e1 in xe)[len(xe) // 2] * 1e3], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(x, op=dist.ReduceOp.MAX)
            return round(float(x), 1)
        # the communicator is set up by one eager step + exchange OUTSIDE any capture; the exchange alone is then timed with
        # events on the launch stream (max over ranks): what one all-reduce of the flat gradient costs when nothing hides it
        model.train_step(a, v, t, y, **sc)
        x_us = {"allreduce": exchange_us("allreduce"), "rs_ag": exchange_us("rs_ag")}
        comm.algo = "allreduce"
        flat_elems = int(model.flat_grad().numel())
        dp = {"nranks": dist.get_world_size(),
              "backend": f"{dist.get_backend()} for bootstrap / barriers; exchange through {'the C-ABI communicator (mmdeer_comm_*)' if comm.backend == 'rccl' else 'torch.distributed'}",
              "payload": args.grad_comm, "payload_bytes": flat_elems * (2 if args.grad_comm == "bf16" else 4),
              "exchange_us": x_us["allreduce"], "exchange_us_by_algo": x_us,
              "exchange_us_note": "median over 10 host-enqueued exchanges (cast + collective(s) + cast back) timed with HIP events on "
                                  "the launch stream, max over ranks: includes the host latency of the collective call; allreduce = one "
                                  "all-reduce, rs_ag = reduce-scatter + all-gather"}
    if not args.eager:
        if comm and os.environ.get("MMDEER_GRAPH_COMM", "1") == "1":
            # The exchange is captured into the step's graph (enqueued from the host after every replay it cost ~85 us per
            # step on a 1-rank group, inside the graph ~25 us).  Two plans: "in-graph" = ONE all-reduce after the pass;
            # "overlapped in-graph" = the backward pass in two calls with the all-reduce of buckets 0-1 (89 % of the
            # gradient) on a side stream under the audio-visual part (splitting the weight-gradient launch costs ~45 us of
            # compute per step, so it only pays when the exposed exchange is longer than that).  With more than one rank
            # BOTH are captured and timed over a few steps and the faster one is kept (MMDEER_DP_OVERLAP=0 / 1 forces
            # one); a plan whose capture fails is skipped, and without any the exchange is enqueued from the host.
            force_plan = os.environ.get("MMDEER_DP_OVERLAP")
            force_algo = os.environ.get("MMDEER_DP_ALGO")          # allreduce | rs_ag: keeps only that single-exchange plan
            cands = []
            if force_plan != "1":
                if force_algo in (None, "allreduce"):
                    cands.append(("in-graph", dict(after=exchange), "allreduce"))
                if force_algo in (None, "rs_ag"):
                    cands.append(("in-graph reduce-scatter + all-gather", dict(after=exchange), "rs_ag"))
            if force_plan == "1" or (force_plan is None and world > 1):
                cands.append(("overlapped in-graph", dict(comm=comm), "allreduce"))
            plans, plan_ms, plan_algo = {}, {}, {}
            for name, kw, algo in cands:
                try:
                    comm.algo = algo            # frozen into the capture (the exchange closure reads it while capturing)
                    plans[name] = model.capture_train_step(a, v, t, y, events=ev, **kw, **sc)
                    plan_algo[name] = algo
                except Exception as e:               # noqa: BLE001
                    print(f"[bench] {name} exchange not capturable here ({type(e).__name__}: {e})", file=sys.stderr)
                    torch.cuda.synchronize()
            comm.algo = "allreduce"
            try:                                                                       # the step without any exchange
                compute_only = model.capture_train_step(a, v, t, y, events=ev, **sc)
            except Exception as e:                   # noqa: BLE001  (this capture was the one outside any guard)
                print(f"[bench] the step without exchange could not be captured ({type(e).__name__}: {e}); host-enqueued steps", file=sys.stderr)
                torch.cuda.synchronize()
                compute_only, plans = None, {}
                dp["why"] = f"graph capture failed: {type(e).__name__}: {e}"
            for name, r in plans.items():
                r(); r()
                plan_ms[name] = round(timed(r, 10), 4)
            compute_ms = None
            if compute_only is not None:
                compute_only(); compute_only()
                compute_ms = round(timed(compute_only, 10), 4)
            dp["compute_only_ms"] = compute_ms
            dp["plan_ms"] = plan_ms
            if plans:
                best = min(plan_ms, key=plan_ms.get)
                replay, comm_in_graph, comm_mode = plans[best], True, best
                comm.algo = plan_algo[best]
                dp["plan"] = best
                dp["exposed_exchange_us"] = round((plan_ms[best] - compute_ms) * 1e3, 1)
                dp["why"] = (f"forced by MMDEER_DP_OVERLAP={force_plan}" if force_plan is not None else
                             ("only plan tried on one rank" if len(plans) == 1 and world == 1 else
                              "fastest of the plans timed over 10 steps: " + ", ".join(f"{k} {v} ms" for k, v in plan_ms.items())))
            else:
                replay, comm_mode = compute_only, "host-enqueued"
                comm.algo = min(x_us, key=x_us.get)
                dp["plan"] = f"host-enqueued ({comm.algo})"
                dp.setdefault("why", "no plan could be captured into the graph on this stack")
        if replay is None and not (comm and os.environ.get("MMDEER_GRAPH_COMM", "1") == "1"):
            replay = model.capture_train_step(a, v, t, y, events=ev, **sc)
            comm_mode = "host-enqueued" if comm else "none"
            if dp is not None and "plan" not in dp:
                dp["plan"], dp["why"] = "host-enqueued", "MMDEER_GRAPH_COMM=0"
    elif comm:
        comm_mode = "host-enqueued"
        dp["plan"], dp["why"] = "host-enqueued", "--eager"

    def one_step(i=None, optimize=False):
        # The metric is forward + loss + backward.  The packed bf16 / transposed weight copies the kernels read are
        # produced by the optimiser step (mmdeer_adamw_step writes them while it updates the fp32 parameters), which
        # the metric excludes; that step is timed separately below (optimizer_ms).
        if replay is not None:
            ld = replay()
        else:
            ld = model.train_step(a, v, t, y, events=ev, prof_events=prof[i] if i is not None else None, **sc)
        if comm and not (comm_in_graph and replay is not None):
            comm.launch(model.flat_grad())
            comm.wait()
        if optimize:
            opt.step()
        return ld

    for _ in range(W):
        one_step(optimize=True)     # warm-up runs real training steps: the timed steps start from optimiser-packed weights
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        ld = one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    own_elapsed = elapsed = time.perf_counter() - t0
    per_rank_ms = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank_ms = [round(float(x) / K * 1e3, 4) for x in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss = float(ld["total_loss"])
    assert loss == loss, "loss is NaN"

    # ---- spread of the step time: every replay between its own pair of events (a separate pass: event records inside the
    #      timed region would perturb the metric).  min / median / max over K steps.
    se = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    torch.cuda.synchronize()
    se[0].record()
    for i in range(K):
        one_step()
        se[i + 1].record()
    torch.cuda.synchronize()
    step_ms = sorted(se[i].elapsed_time(se[i + 1]) for i in range(K))

    # ---- every launch of the step, measured in THIS run: the library's launch trace (mmdeer_trace_begin / _end) records an event
    #      behind every launch of eager steps on the launch stream; consecutive events = that launch (kernel + the gap to the next
    #      enqueue: the GPU runs these launches back to back while the host stays ahead; a replayed graph has no gaps at all)
    launch_us = None
    try:
        import ctypes as C
        lib_t = _lib.load()
        NE, NS = 32, 20
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(NE + 1)] for _ in range(NS)]
        for row in evs:
            for x in row:
                x.record()
        torch.cuda.synchronize()
        labels, counts = None, []
        for i in range(NS):
            evs[i][0].record()
            arr = (C.c_void_p * NE)(*[x.cuda_event for x in evs[i][1:]])
            _lib.check(lib_t.mmdeer_trace_begin(arr, NE))
            model.train_step(a, v, t, y, **sc)
            n = lib_t.mmdeer_trace_end()
            counts.append(n)
            if labels is None:
                labels = [lib_t.mmdeer_trace_label(k).decode() for k in range(n)]
        torch.cuda.synchronize()
        if labels and all(c == len(labels) for c in counts):
            launch_us = {}
            for k, name in enumerate(labels):
                d = sorted(evs[i][k].elapsed_time(evs[i][k + 1]) * 1e3 for i in range(2, NS))      # the first steps warm the eager path up
                launch_us[f"{k:02d} {name}"] = round(d[len(d) // 2], 2)
    except Exception as e:               # noqa: BLE001  (diagnostics only)
        print(f"[bench] launch trace failed ({type(e).__name__}: {e})", file=sys.stderr)

    # ---- the roofline kernel, measured in THIS run.  bf16 fused plan: N back-to-back launches of tri_fused_kernel<0> on the
    #      step's own operands (xtok / obar / probs of the workspace, the packed head-major weight image and the fp32 bias of
    #      the weights buffer) captured into one HIP graph; events around a replay / N.  Replayed launches run gap-free (the
    #      rocprofv3 trace of a replayed graph shows every kernel starting where the previous one ended), so this is the
    #      same per-launch duration rocprofv3 --stats averages.  Other plans: HIP events around the launch in eager steps.
    fused = args.dtype == "bf16" and _lib.get_option("fused_attn") == 1
    roof_us, roof_how = None, None
    if fused:
        lib = _lib.load()
        ws, wb = model._workspace(B, dev), model._weights(dev)
        wo = lambda n: ws.data_ptr() + lib.mmdeer_workspace_offset(B, 0, n.encode())
        bias_off = next(lib.mmdeer_param_offset(i) for i in range(lib.mmdeer_num_params())
                        if lib.mmdeer_param_name(i).decode() == "fusion.trimodal_fusion.modality_attention.in_proj_bias")
        whm = wb.data_ptr() + lib.mmdeer_weights_offset(0, b"wqkv_hm")
        bias = wb.data_ptr() + lib.mmdeer_weights_offset(0, b"vpack") + 4 * bias_off
        NL, NR, NX = 20, 10, 10
        # In the step X (xtok) was written by the two launches before and is NOT resident in the L2 of the XCD that reads it;
        # launched back to back on ONE buffer the kernel would find its operands in L2 from the previous launch (measured
        # 13.3 us against 14.9 us in the step).  The launches therefore rotate over NX copies of the step's X (84 MB in all:
        # beyond the 32 MB of L2, inside the Infinity Cache, as in the step).
        nx = 2 * B * 512 * 2
        xoff = lib.mmdeer_workspace_offset(B, 0, b"xtok")
        xcopies = ws[xoff:xoff + nx].clone().repeat(NX).contiguous()

        def launch_roof(i=0):
            _lib.check(lib.mmdeer_trimodal_fused_fwd(xcopies.data_ptr() + (i % NX) * nx, whm, bias, wo("obar"), wo("probs"), None, None, None,
                                                     B, 1, 0.3, model.dropout_seed, int(model._step), _lib.current_stream()))
        launch_roof()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(NL):
                launch_roof(i)
        g.replay()
        rev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(NR)]
        for e0, e1 in rev:
            e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        gemm_ms = sorted(e0.elapsed_time(e1) / NL for e0, e1 in rev)
        roof_how = (f"HIP events around a HIP graph of {NL} back-to-back launches of the kernel on the step's own operands "
                    f"(X rotated over {NX} copies so that it is not L2-resident from the previous launch, as in the step), / {NL}; "
                    f"{NR} replays (this run; no profiler attached)")
    else:
        for i in range(K):
            model.train_step(a, v, t, y, prof_events=prof[i])
        torch.cuda.synchronize()
        gemm_ms = sorted(e0.elapsed_time(e1) for e0, e1 in prof)
        roof_how = "HIP events around the kernel's launch inside K eager steps on the launch stream (includes the event pair and the launch gap)"
    # the same K steps again with the optimiser step inside (reported beside the metric, never as `value`)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(K):
        one_step(optimize=True)
    torch.cuda.synchronize()
    full_elapsed = time.perf_counter() - t1

    if rank == 0:
        avg_ms = sum(gemm_ms) / len(gemm_ms)
        flops = 2.0 * (2 * B) * 512 * 1536                      # algorithmic: SURVEY 8d, 3.146 MFLOP/sample forward
        step_flops = 3 * 2.0 * 3950336 * B                       # 7.90 MFLOP/sample forward, x3 for the train step
        peak = BF16_MFMA_PEAK if args.dtype == "bf16" else F32_MFMA_PEAK
        achieved = flops / (avg_ms * 1e-3)
        traffic, traffic_src = measured_traffic(args.dtype, B)
        out = {
            "metric": "samples/sec fwd+bwd at B=4096 (84/256/768-dim)",
            "value": round(world * B * K / elapsed, 1),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "ms_per_step_min": round(step_ms[0], 4), "ms_per_step_median": round(step_ms[len(step_ms) // 2], 4),
            "ms_per_step_max": round(step_ms[-1], 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"configs[2]: fusion+DEER forward + MultiTaskDEERLoss + backward (dropout 0.3, gradients "
                                   f"materialised in the flat buffer), B={B}/GPU, (B,84)+(B,256)+(B,768) {args.dtype} feature "
                                   f"blocks, random-init weights; optimiser step excluded as the metric defines "
                                   f"(it maintains the packed weight copies and is timed separately)", "global_batch": world * B,
                       "parallelism": (f"dp{world} (one process per GPU, gradient all-reduce over RCCL, {args.grad_comm} payload, "
                                        f"{comm_mode})" if world > 1 else "single")},
            "inputs": "one resident batch replayed: the same (B,84)+(B,256)+(B,768) blocks every step (9 MB at B=4096 bf16, Infinity-Cache "
                      "resident), fresh dropout masks per step; host-to-device copies are outside the metric (DESIGN.md section 7)",
            "launch": "eager" if replay is None else "hip-graph replay",
            # which launch plan ran: chosen by timing the plans on this GPU (untimed set-up), or the library's current options
            "launch_plan": launch_plan if launch_plan is not None else
                           {"plan": "as set", "options": {k: _lib.get_option(k) for k in ("chain", "chain_bwd", "chain_nig")},
                            "why": "--no-autotune / --eager / MMDEER_CHAIN* given"},
            "grad_exchange": comm_mode + (", exact-global loss statistics" if exact else ""),
            # per-launch durations of eager steps of this run (median over 18 steps, HIP events behind every launch: kernel + the gap
            # to the next one; the replayed graph the metric times has no gaps)
            "launch_us": launch_us,
            "final_loss": round(loss, 6),
            "train_step_with_optimizer_ms": round(full_elapsed / K * 1e3, 4),     # fwd + bwd + clip + AdamW + weight pack
            "optimizer_ms": round((full_elapsed - own_elapsed) / K * 1e3, 4),
            "roofline": {"bound": "mfma",
                         "kernel": ("tri_fused_kernel<0> (trimodal in_proj M=2B K=512 N=1536 + 2-token attention fused: q|k|v stay on chip)" if fused
                                    else ("gemm_nt256_kernel<192>" if args.dtype == "bf16" else "gemm_group_kernel") + " (trimodal in_proj, M=2B K=512 N=1536)"),
                         "achieved": round(achieved / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         # HBM-side bytes of one launch from separate rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE,
                         # tools/gpu_pmc_fused.sh -> profiles/pmc_traffic.json); null unless taken on exactly these kernel sources
                         "traffic": traffic, "traffic_source": traffic_src,
                         # fused: X 8.39 + W 1.57 + obar 4.19 + probs 0.52 MB; unfused GEMM: A 8.39 + W 1.57 + C 25.17 MB
                         "algorithmic_bytes": (2 * B * 512 * 2 + 1536 * 512 * 2 + B * 512 * 2 + B * 32 * 4) if fused
                                              else (2 * B * 512 * 2 + 1536 * 512 * 2 + 2 * B * 1536 * 2),
                         "avg_launch_us": round(avg_ms * 1e3, 2), "median_launch_us": round(gemm_ms[len(gemm_ms) // 2] * 1e3, 2),
                         "min_launch_us": round(gemm_ms[0] * 1e3, 2), "timing": roof_how,
                         # the whole step against the same peak: 97.1 GFLOP (algorithmic) / step time
                         "step_frac": round(step_flops / (elapsed / K) / peak, 4)},
        }
        if dp is not None:
            if per_rank_ms is not None:
                dp["per_rank_ms_per_step"] = per_rank_ms
            out["data_parallel"] = dp
        rp = recorded_profile(B, args.dtype)
        if rp is not None:
            out["recorded_profile"] = rp     # read from committed files, NOT measured by this run
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B)
        print(json.dumps(out), flush=True)
    if world > 1 or force_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
