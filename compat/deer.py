"""`from deer import test_deer_implementation` (run_multimodal_deer.py:76; src/models/deer.py).  DEERLoss here is the
reference's loss variant 1 (deer.py:111-195); the DEER head itself is part of mmdeer.model.MultimodalDEER."""
from mmdeer.losses import DEERLossV1 as DEERLoss  # noqa: F401
from mmdeer.side import CrossModalAttention, ModalityEncoders  # noqa: F401


def test_deer_implementation() -> bool:
    """The reference's self-test (deer.py:428-470: build the head, run it, check shapes and the loss) on the HIP path: one
    forward + MultiTaskDEERLoss + backward of MultimodalDEER on cuda:0.  Raises without a GPU (there is no CPU fallback)."""
    import torch

    from mmdeer import synth
    from mmdeer.model import ModelConfig, MultimodalDEER
    if not torch.cuda.is_available():
        raise RuntimeError("test_deer_implementation: the mmdeer hot path needs an MI355X (no CPU fallback)")
    m = MultimodalDEER(ModelConfig(compute_dtype="fp32")).to("cuda:0").train()
    b = {k: torch.from_numpy(v).to("cuda:0") for k, v in synth.make_batch(8).items()}
    out = m(b["audio"], b["video"], b["text"])
    assert out["mu_all"].shape == (8, 3) and all(out[f"{d}_alpha"].min() > 1 for d in ("valence", "arousal", "dominance"))
    loss = m.compute_loss(out, b["targets"])
    loss["total_loss"].backward()
    assert torch.isfinite(loss["total_loss"]) and all(p.grad is not None for p in m.live_parameters())
    print("DEER implementation test passed (mmdeer HIP path)")
    return True


test_deer_implementation.__test__ = False      # a function of the reference's API, not a pytest case
