#!/usr/bin/env python3
"""Launcher counterpart of the reference's experiments/run_multimodal_deer.py (the reference script cannot
run as shipped; this reproduces the contract it encodes):

    python experiments/run_multimodal_deer.py --mode full --quick --batch_size 32

config -> model (ModelConfig / CompleteDEERModel alias) -> synthetic TensorDataset loaders with the script's
recipe (:329-351) -> DEERTrainer.train -> evaluate -> checkpoint + JSON report.  Needs a GPU: the model has no
CPU path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
from torch.utils.data import DataLoader, TensorDataset  # noqa: E402

from mmdeer import synth  # noqa: E402
from mmdeer import stackb  # noqa: E402
from mmdeer.model import CompleteDEERModel, ModelConfig  # noqa: E402
from mmdeer.trainer import DEERTrainer, TrainingConfig  # noqa: E402

DEFAULT_CONFIG = {  # run_multimodal_deer.py:163-188
    "model": {"audio_dim": 84, "video_dim": 256, "text_dim": 768, "fusion_dim": 512, "emotion_dims": 3,
              "dropout": 0.3, "attention_heads": 8},
    "training": {"learning_rate": 1e-4, "batch_size": 32, "num_epochs": 50, "weight_decay": 1e-5, "gradient_clip": 1.0},
}


def synthetic_loader(n, batch_size, split, seed):
    b = synth.make_batch(n, seed=seed)
    ds = TensorDataset(*(torch.from_numpy(b[k]) for k in ("audio", "video", "text", "targets")))
    return {f"synthetic_{split}": DataLoader(ds, batch_size=batch_size, shuffle=(split == "train"))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="full", choices=["full", "train", "evaluate", "test"])
    ap.add_argument("--config", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--learning_rate", type=float, default=None)
    ap.add_argument("--output_dir", default="./results")
    ap.add_argument("--compute_dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--graph", action="store_true", help="replay the training step as a captured HIP graph (TrainingConfig.use_graph: "
                    "one graph per batch shape, the ragged tail batch runs eagerly); GPU only")
    ap.add_argument("--stack", default="c", choices=["c", "b"],
                    help="which class the script's name CompleteDEERModel resolves to: c = multimodal_deer.MultimodalDEER (the model "
                         "the reference trains, fused HIP step), b = complete_project.CompleteDEERModel (operator-sequence training)")
    args = ap.parse_args()

    cfg = json.loads(json.dumps(DEFAULT_CONFIG))
    if args.config and os.path.exists(args.config):
        user = yaml.safe_load(open(args.config)) or {}
        for k in ("model", "training"):
            cfg[k].update(user.get(k, {}) or {})
    if args.quick:                       # run_multimodal_deer.py:854-859
        cfg["training"]["num_epochs"], cfg["training"]["batch_size"] = 5, 8
    for arg, key in (("batch_size", "batch_size"), ("epochs", "num_epochs"), ("learning_rate", "learning_rate")):
        if getattr(args, arg) is not None:
            cfg["training"][key] = getattr(args, arg)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if not torch.cuda.is_available():
        sys.exit("run_multimodal_deer: a ROCm GPU is required (the fusion + DEER path has no CPU implementation)")
    device = torch.device("cuda:0")
    exp_dir = os.path.join(args.output_dir, time.strftime("experiment_%Y%m%d_%H%M%S"))
    os.makedirs(exp_dir, exist_ok=True)

    if args.stack == "b":
        fields = stackb.ModelConfig.__dataclass_fields__
        model = stackb.CompleteDEERModel(stackb.ModelConfig(**{k: v for k, v in cfg["model"].items() if k in fields}, dropout_seed=args.seed),
                                         compute_dtype=args.compute_dtype).to(device)
    else:
        mc = ModelConfig(**{k: v for k, v in cfg["model"].items() if k in ModelConfig.__dataclass_fields__},
                         compute_dtype=args.compute_dtype, seed=args.seed)
        model = CompleteDEERModel(mc).to(device)
    print(f"model: {sum(p.numel() for p in model.parameters()):,} parameters, compute {args.compute_dtype}")
    bs = cfg["training"]["batch_size"]
    train = synthetic_loader(1000, bs, "train", args.seed)
    val = synthetic_loader(200, bs, "val", args.seed + 1)
    test = synthetic_loader(200, bs, "test", args.seed + 2)
    tc = TrainingConfig(learning_rate=cfg["training"]["learning_rate"], batch_size=bs, num_epochs=cfg["training"]["num_epochs"],
                        weight_decay=cfg["training"]["weight_decay"], gradient_clip=cfg["training"]["gradient_clip"],
                        output_dir=os.path.join(exp_dir, "models"), log_dir=os.path.join(exp_dir, "logs"),
                        checkpoint_dir=os.path.join(exp_dir, "checkpoints"), use_graph=bool(args.graph))
    trainer = DEERTrainer(model, tc, device)
    report = {"config": cfg, "mode": args.mode}
    if args.mode in ("full", "train"):
        t0 = time.time()
        hist = trainer.train(train, val)
        report["training_time"] = time.time() - t0
        report["history"] = hist
        trainer.save_checkpoint(os.path.join(exp_dir, "models", "final_model.pt"), report["training_time"])
        print(f"trained {len(hist['train_loss'])} epochs: loss {hist['train_loss'][0]:.4f} -> {hist['train_loss'][-1]:.4f}")
    if args.mode in ("full", "evaluate", "test"):
        report["evaluation"] = trainer.evaluate_model(test)
        print("evaluation:", {k: round(v, 4) for k, v in report["evaluation"].items()})
        model.eval()
        with torch.no_grad():
            b = {k: torch.from_numpy(v).to(device) for k, v in synth.make_batch(4, seed=7).items()}
            out = (model(b["audio"], b["video"], b["text"]) if args.stack == "b" else
                   model({"audio": b["audio"], "video": b["video"], "text": b["text"]}))     # :707-719
            preds, unc = model.get_predictions_and_uncertainties(out)
        report["sample_predictions"] = {"predictions": preds.cpu().tolist(), "uncertainties": unc.cpu().tolist(),
                                        "nig_keys": [k for k in (("valence_mu", "valence_nu") if args.stack == "b" else ("gamma", "nu", "alpha", "beta")) if k in out]}
    with open(os.path.join(exp_dir, "report.json"), "w") as f:
        json.dump(report, f, indent=2, default=float)
    print("report:", os.path.join(exp_dir, "report.json"))


if __name__ == "__main__":
    main()
