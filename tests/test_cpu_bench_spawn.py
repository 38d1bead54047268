"""`python bench.py --gpus N` must start N ranks itself: exercised here on the CPU through the
bench's --plumbing mode (gloo rendezvous, same barrier / MAX-over-ranks bracket, one JSON line from rank 0)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                          timeout=timeout)


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--plumbing"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["plumbing"] is True and d["steps"] == 3
    # the diagnostics a first real multi-GPU run needs to be read without a second run: the same keys
    # as the GPU line's `data_parallel` block
    dp = d["data_parallel"]
    assert dp["nranks"] == 2 and dp["backend"] == "gloo"
    assert dp["payload_bytes"] == 4 * 2909120 or dp["payload_bytes"] % 256 == 0      # the flat gradient buffer, fp32
    assert dp["exchange_us"] > 0 and dp["plan"] == "host-enqueued" and dp["why"]
    assert len(dp["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in dp["per_rank_ms_per_step"])
    assert max(dp["per_rank_ms_per_step"]) <= d["ms_per_step"] * 1.5 + 1.0


def test_world_size_mismatch_fails_loudly():
    # a launcher that started ONE rank for --gpus 2 must not yield an n_gpus=1 line
    r = _run(["--gpus", "2", "--plumbing"], env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_more_gpus_than_visible_fails_before_any_rank_starts():
    r = _run(["--gpus", "2", "--steps", "1"])          # no GPU in the build container; 1 on a gpurun box
    import torch
    if torch.cuda.device_count() >= 2:
        return
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr
