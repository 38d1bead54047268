"""Thin trainer / evaluator reproducing the hook contract of the reference's DEERTrainer
(src/training/training.py:75-507) on top of the fused HIP step.

Kept: TrainingConfig field names (:38-72); AdamW eps 1e-8 with three parameter groups, 'encoder'-named
parameters at 0.5 x lr (:121-150); cosine / plateau / exponential schedules (:152-174); global-norm clip 1.0
(:219-222); dataset weighting of the loss (:211-212); both batch formats (dict with *_features/targets, and the
4-tuple of the script's TensorDataset, run_multimodal_deer.py:342, 414-415); JSON-serialisable history.
Not reproduced (out of scope): curriculum sampling, TensorBoard.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import asdict, dataclass, field
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from .metrics import StreamingMetrics, validation_metrics
from .model import MultimodalDEER
from .optim import FlatAdamW, FusedAdamW


@dataclass
class TrainingConfig:
    learning_rate: float = 1e-4
    weight_decay: float = 1e-5
    gradient_clip: float = 1.0
    batch_size: int = 32
    num_epochs: int = 100
    scheduler_type: str = "cosine"
    fused_optimizer: bool = True   # clip + AdamW + weight pack on the device (optim.FusedAdamW); False: torch.optim.AdamW
    use_graph: bool = False        # replay train_step (Stack B: train_step_fused) as a captured HIP graph for full batches of one shape (needs fused_optimizer)
    warmup_epochs: int = 5
    patience: int = 10
    evidence_weight: float = 1.0
    kl_weight: float = 0.1
    attention_reg_weight: float = 0.1
    dataset_weights: Dict[str, float] = field(default_factory=lambda: {"iemocap": 1.0, "ravdess": 0.8, "meld": 0.6})
    curriculum_learning: bool = True
    val_frequency: int = 5
    save_frequency: int = 10
    early_stopping: bool = True
    output_dir: str = "./results"
    log_dir: str = "./logs"
    checkpoint_dir: str = "./checkpoints"


def unpack_batch(batch, device):
    """dict batches (training.py:201-204) or (audio, video, text, emotions) tuples (run_multimodal_deer.py:414-415)."""
    if isinstance(batch, dict):
        a, v, t, y = (batch[k] for k in ("audio_features", "video_features", "text_features", "targets"))
    else:
        a, v, t, y = batch
    nb = device.type == "cuda"
    return tuple(x.to(device, non_blocking=nb) for x in (a, v, t, y))


class DEERTrainer:
    def __init__(self, model: MultimodalDEER, config: Optional[TrainingConfig] = None, device=None, comm=None):
        self.config = config or TrainingConfig()
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        self.model = model.to(self.device)
        self.comm = comm                     # optional mmdeer.parallel.BucketedAllReduce (data parallel)
        # Data parallel: the ranks hold identical parameters (same `seed`) but must not draw identical dropout masks for their
        # local rows -- the mask hash is a function of (dropout_seed, step, site, LOCAL row, column).  Unless the caller chose
        # a dropout_seed, rank r gets seed + 1000003 r (what bench.py does for its ranks).
        world = int(getattr(comm, "world", 1) or 1) if comm is not None else 1
        if world > 1 and hasattr(model, "config") and getattr(model.config, "dropout_seed", None) in (None, 0):
            import torch.distributed as dist
            rank = int(getattr(comm, "rank", dist.get_rank(getattr(comm, "group", None)) if dist.is_initialized() else 0))
            if getattr(model.config, "dropout_seed", None) is None or rank:      # (Stack B's ModelConfig defaults to 0: rank 0 keeps it)
                model.config.dropout_seed = int(getattr(model.config, "seed", 0)) + 1000003 * rank
        # Stack B (stackb.CompleteDEERModel) trains through autograd: forward -> compute_loss -> backward into .grad, then
        # clip_grad_norm_ + torch.optim.AdamW exactly as training.py:205-224; Stack C has the fused step + flat buffer
        self.generic = not hasattr(model, "flat_grad")
        # ... unless it offers the fused step (stackb.CompleteDEERModel.train_step_fused: the same operator sequence without
        # autograd, gradients in a flat buffer) and the fused optimiser is wanted: 0.94 ms against 2.95 ms per step at B = 4096
        self.fused_b = self.generic and hasattr(model, "train_step_fused") and bool(self.config.fused_optimizer) and self.device.type == "cuda"
        # data parallel: the exchange is one call on the flat gradient buffer between the step and the optimiser -- Stack C's, and the
        # fused Stack B step's (its flat buffer: CompleteDEERModel._flat); the autograd route of Stack B has no flat buffer
        if self.generic and comm is not None and not self.fused_b:
            raise NotImplementedError("data-parallel training needs a flat gradient buffer: MultimodalDEER (Stack C), or Stack B with "
                                      "fused_optimizer on a GPU")
        if self.fused_b and getattr(comm, "exact_global", False):
            raise NotImplementedError("the exact-global loss mode exchanges Stack C's loss statistics inside the step; Stack B trains with DDP semantics")
        self.optimizer = self._create_optimizer()
        self.scheduler = self._create_scheduler()
        self.current_epoch = 0
        self._resumed = False                # load_checkpoint() sets it: train() then continues after `current_epoch`
        self.history = {"train_loss": [], "val_loss": [], "train_ccc": [], "val_ccc": [], "learning_rate": [], "grad_norm": []}
        for d in (self.config.output_dir, self.config.log_dir, self.config.checkpoint_dir):
            os.makedirs(d, exist_ok=True)

    # ---- optimiser semantics of training.py:121-174
    def _create_optimizer(self):
        enc, att, rest = [], [], []
        for name, p in self.model.named_parameters():
            (enc if "encoder" in name else att if "attention" in name else rest).append(p)
        lr = self.config.learning_rate
        groups = [g for g in ({"params": enc, "lr": lr * 0.5}, {"params": att, "lr": lr}, {"params": rest, "lr": lr}) if g["params"]]
        if self.fused_b:
            self.model._flat(self.device)          # the parameters move into the flat buffer BEFORE the groups capture them
            return FlatAdamW(self.model, lr=lr, weight_decay=self.config.weight_decay, eps=1e-8, max_grad_norm=self.config.gradient_clip)
        if self.config.fused_optimizer and not self.generic:
            # clip_grad_norm_ + AdamW + weight pack as one device-side step (optim.FusedAdamW)
            return FusedAdamW(self.model, groups, lr=lr, weight_decay=self.config.weight_decay, eps=1e-8,
                              max_grad_norm=self.config.gradient_clip)
        return torch.optim.AdamW(groups, weight_decay=self.config.weight_decay, eps=1e-8)

    def _create_scheduler(self):
        c = self.config
        if c.scheduler_type == "cosine":
            return torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=c.num_epochs, eta_min=1e-6)
        if c.scheduler_type == "plateau":
            return torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", patience=max(1, c.patience // 2), factor=0.5)
        return torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=0.95)

    def clip_gradients(self) -> torch.Tensor:
        """clip_grad_norm_(model.parameters(), gradient_clip) on the flat gradient buffer (its alignment gaps are
        zeros, and unused parameters have no gradient, so its 2-norm IS the global norm).  No host sync."""
        flat = self.model.flat_grad()
        total = torch.linalg.vector_norm(flat)
        flat.mul_(torch.clamp(self.config.gradient_clip / (total + 1e-6), max=1.0))
        return total

    # ---- optional HIP-graph replay of the step (config.use_graph): one graph per (batch size, dtype), fed through
    #      static input tensors; batches of another shape (the last, ragged one) take the eager path
    def _graph_ok(self, a: torch.Tensor) -> bool:
        return bool(self.config.use_graph) and isinstance(self.optimizer, FusedAdamW) and a.is_cuda

    def _graph_step(self, a, v, t, y):
        key = (a.shape[0], a.dtype)
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        entry = self._graphs.get(key)
        if entry is None:
            if self._graphs and key[0] < max(k[0] for k in self._graphs):
                return None                      # a smaller tail batch: not worth a capture
            static = tuple(x.clone() for x in (a, v, t, y.float()))
            # exact-global mode: the statistics exchange sits between forward and backward INSIDE the captured step (the
            # eager fallback below passes the same communicator); without it a replayed step would be a per-shard step
            # whose gradients the SUM exchange then scales by the world size
            sc = self.comm if getattr(self.comm, "exact_global", False) else None
            replay = self.model.capture_train_step(*static, stats_comm=sc)
            entry = self._graphs[key] = (static, replay)
            return replay.first              # the capture's eager warm-up WAS this batch's step: gradients are in place
        static, replay = entry
        for dst, src in zip(static, (a, v, t, y)):
            dst.copy_(src, non_blocking=True)
        return replay()

    def _graph_step_b(self, a, v, t, y):
        """Stack B (fused step): one captured graph per (batch size, dtype); the capture's single eager warm-up IS the first batch's
        step (same dropout step as the eager trainer would have used), the replays continue the mask stream."""
        key = (a.shape[0], a.dtype)
        if not hasattr(self, "_graphs_b"):
            self._graphs_b = {}
        replay = self._graphs_b.get(key)
        if replay is None:
            if self._graphs_b and key[0] < max(k[0] for k in self._graphs_b):
                return None                      # a smaller tail batch: not worth a capture
            replay = self._graphs_b[key] = self.model.capture_train_step_fused(a, v, t, y, warmup=1)
            return replay.first
        return replay(a, v, t, y)

    # ---- one epoch (training.py:176-245)
    def train_epoch(self, train_loaders: Dict[str, Iterable]) -> Dict[str, float]:
        self.model.train()
        keys = ("total_loss", "deer_loss", "nll_loss", "evidence_reg", "kl_reg")
        sums = torch.zeros(len(keys), device=self.device, dtype=torch.float64)
        total, norms = 0, []
        events = self.comm.events if self.comm is not None else None
        for name, loader in train_loaders.items():
            w = float(self.config.dataset_weights.get(name, 1.0))
            for batch in loader:
                a, v, t, y = unpack_batch(batch, self.device)
                if self.fused_b:
                    ld = self._graph_step_b(a, v, t, y) if (self.config.use_graph and a.is_cuda) else None
                    if ld is None:
                        ld = self.model.train_step_fused(a, v, t, y)
                    if self.comm is not None:            # mean of the ranks' gradients (DDP semantics), in place on the flat buffer
                        flat_g = self.model._flat(self.device)["g"]
                        self.comm.launch(flat_g)
                        self.comm.wait(flat_g)
                    norms.append(self.optimizer.step(grad_scale=w))    # weighted_loss = total_loss * weight (:211-212) as a gradient scale
                    bs = a.shape[0]
                    sums += torch.stack([ld[k].double() for k in keys]) * bs
                    total += bs
                    continue
                if self.generic:
                    self.optimizer.zero_grad(set_to_none=True)
                    ld = self.model.compute_loss(self.model(a, v, t), y)
                    (ld["total_loss"] * w).backward()                       # weighted_loss.backward() (:211-216)
                    norms.append(torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.config.gradient_clip))
                    self.optimizer.step()
                    bs = a.shape[0]
                    sums += torch.stack([ld[k].detach().double() for k in keys]) * bs
                    total += bs
                    continue
                # no zero_grad(): the fused step overwrites every live gradient slice of the flat buffer
                ld = self._graph_step(a, v, t, y) if self._graph_ok(a) else None
                if ld is None:
                    # forward + loss + backward, fused; a communicator in exact-global mode also sums the loss statistics
                    sc = self.comm if getattr(self.comm, "exact_global", False) else None
                    ld = self.model.train_step(a, v, t, y, events=events, stats_comm=sc)
                if self.comm is not None:
                    self.comm.launch(self.model.flat_grad())
                    self.comm.wait(self.model.flat_grad())
                if isinstance(self.optimizer, FusedAdamW):
                    # weighted_loss = total_loss * weight (:211-212) enters as a gradient scale; clip + step on the device
                    norms.append(self.optimizer.step(grad_scale=w))
                else:
                    if w != 1.0:
                        self.model.flat_grad().mul_(w)
                    norms.append(self.clip_gradients())
                    self.optimizer.step()
                bs = a.shape[0]
                sums += torch.stack([ld[k].double() for k in keys]) * bs    # accumulated on device: no .item() per batch
                total += bs
        out = {k: float(s) / max(total, 1) for k, s in zip(keys, sums.cpu())}
        out["attention_reg"] = 0.0
        out["grad_norm"] = float(torch.stack(norms).mean()) if norms else 0.0
        return out

    def validate_epoch(self, val_loaders: Dict[str, Iterable]) -> Dict[str, float]:
        return evaluate_loaders(self.model, val_loaders, self.device)

    def evaluate_model(self, test_loaders: Dict[str, Iterable]) -> Dict[str, float]:
        m = self.validate_epoch(test_loaders)
        m["test_loss"] = m.pop("val_loss")
        return m

    def train(self, train_loaders, val_loaders) -> Dict:
        """The loop of training.py:356-455.  Validation every ``val_frequency`` epochs (:379); the best model is the one with
        the highest ``ccc_overall`` (:406-418) and early stopping counts validations without improvement (:419-425); a
        checkpoint every ``save_frequency`` epochs (:433-439) and a final one (:445-448); the plateau scheduler steps on the
        validation loss when there is one, else on the training loss (:428-429).  Without validation loaders nothing is
        validated, so there is no best model and no early stopping.  Returns a JSON-serialisable history
        (run_multimodal_deer.py:503-509)."""
        t0 = time.time()
        c = self.config
        first = 0
        if self._resumed:          # continue a checkpointed run: next epoch, best model / patience as they were
            first, self._resumed = self.current_epoch + 1, False
            if first >= c.num_epochs:
                import warnings
                warnings.warn(f"DEERTrainer.train: the loaded checkpoint finished epoch {first - 1} and num_epochs is {c.num_epochs}: "
                              "nothing left to train (load_checkpoint(path, resume=False) starts a new run from the loaded weights)")
        else:
            self.best_ccc, self.best_val_loss, self.patience_counter = -float("inf"), float("inf"), 0
        vf, sf = max(1, int(c.val_frequency)), max(1, int(c.save_frequency))
        for epoch in range(first, c.num_epochs):
            self.current_epoch = epoch
            tr = self.train_epoch(train_loaders)
            va, stop = None, False
            if val_loaders and epoch % vf == 0:
                va = self.validate_epoch(val_loaders)
                self.history["train_loss"].append(tr["total_loss"])
                self.history["val_loss"].append(va["val_loss"])
                self.history["train_ccc"].append(float("nan"))
                self.history["val_ccc"].append(va.get("ccc_overall", float("nan")))
                for k in ("ccc_valence", "ccc_arousal", "ccc_dominance", "ece"):
                    self.history.setdefault(k, []).append(va.get(k, 0.0))
                cur = va.get("ccc_overall", 0.0)
                if cur > self.best_ccc:
                    self.best_ccc, self.best_val_loss, self.patience_counter = cur, va["val_loss"], 0
                    self.save_checkpoint(os.path.join(c.checkpoint_dir, "best_model.pt"), time.time() - t0, epoch=epoch, loss=va["val_loss"])
                else:
                    self.patience_counter += 1
                stop = bool(c.early_stopping) and self.patience_counter >= c.patience
            elif not val_loaders:
                self.history["train_loss"].append(tr["total_loss"])
                self.history["val_loss"].append(float("nan"))
                self.history["train_ccc"].append(float("nan"))
                self.history["val_ccc"].append(float("nan"))
            self.history["learning_rate"].append(self.optimizer.param_groups[0]["lr"])
            self.history["grad_norm"].append(tr["grad_norm"])
            if stop:
                break
            if isinstance(self.scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
                self.scheduler.step(va["val_loss"] if va is not None else tr["total_loss"])
            else:
                self.scheduler.step()
            if epoch % sf == 0:
                self.save_checkpoint(os.path.join(c.checkpoint_dir, f"checkpoint_epoch_{epoch}.pt"), time.time() - t0, epoch=epoch,
                                     loss=tr["total_loss"])
        self.history["training_time"] = time.time() - t0
        self.save_checkpoint(os.path.join(c.checkpoint_dir, "final_model.pt"), self.history["training_time"], epoch=self.current_epoch)
        with open(os.path.join(c.output_dir, "training_history.json"), "w") as f:
            json.dump(self.history, f, indent=2)
        return self.history

    def save_checkpoint(self, path: str, training_time: float = 0.0, epoch: Optional[int] = None, loss: Optional[float] = None) -> None:
        """Checkpoint layout of run_multimodal_deer.py:512-517 (model_state_dict / training_config / training_history /
        training_time) plus what a resume needs and the reference's ModelCheckpoint calls pass (training.py:415-418):
        optimiser state (FusedAdamW: step count and the flat moment buffers), scheduler state, epoch, loss, the model's
        dropout stream position (masks are a hash of (seed, step): without it a resumed run would replay the masks of steps
        0..N instead of continuing the stream) and the best-model / early-stopping bookkeeping of ``train``."""
        torch.save({"model_state_dict": self.model.state_dict(), "training_config": asdict(self.config),
                    "training_history": self.history, "training_time": training_time,
                    "optimizer_state_dict": self.optimizer.state_dict(), "scheduler_state_dict": self.scheduler.state_dict(),
                    "epoch": epoch, "loss": loss, "dropout_state": dropout_state(self.model),
                    "trainer_state": {"best_ccc": getattr(self, "best_ccc", -float("inf")),
                                      "best_val_loss": getattr(self, "best_val_loss", float("inf")),
                                      "patience_counter": getattr(self, "patience_counter", 0)}}, path)

    def load_checkpoint(self, path: str, resume: bool = True) -> Dict:
        """Load a ``save_checkpoint`` file: parameters, optimiser moments / step, scheduler, history.  ``resume=True`` (default)
        makes the next ``train()`` CONTINUE the checkpointed run -- epoch + 1, best model and patience as they were; with
        ``resume=False`` (a best-model file loaded for evaluation or fine-tuning) the next ``train()`` runs its ``num_epochs`` from
        epoch 0 like the reference trainer after a load."""
        ck = torch.load(path, map_location=self.device, weights_only=False)
        self.model.load_state_dict(ck["model_state_dict"])
        if ck.get("optimizer_state_dict") is not None:
            self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        if ck.get("scheduler_state_dict") is not None:
            self.scheduler.load_state_dict(ck["scheduler_state_dict"])
        self.history = ck.get("training_history", self.history)
        if ck.get("dropout_state") is not None:
            load_dropout_state(self.model, ck["dropout_state"])
        ts = ck.get("trainer_state") or {}
        self.best_ccc = ts.get("best_ccc", -float("inf"))
        self.best_val_loss = ts.get("best_val_loss", float("inf"))
        self.patience_counter = ts.get("patience_counter", 0)
        if ck.get("epoch") is not None:
            self.current_epoch = int(ck["epoch"])
            self._resumed = bool(resume)     # train() continues with epoch + 1
        return ck


def dropout_state(model) -> Dict[str, int]:
    """Position of the model's dropout stream: Stack C counts training forwards in ``_step``, Stack B in ``_train_step``
    (their device-side counters are re-aligned to it by the captured steps' ``replay``)."""
    return {k: int(getattr(model, k)) for k in ("_step", "_train_step") if hasattr(model, k)}


def load_dropout_state(model, state: Dict[str, int]) -> None:
    for k, v in state.items():
        if hasattr(model, k):
            setattr(model, k, int(v))


@torch.no_grad()
def evaluate_loaders(model, loaders: Dict[str, Iterable], device=None) -> Dict[str, float]:
    """Validation pass of training.py:247-314 for any model of this package that maps (audio, video, text) to an output
    dictionary understood by its ``get_predictions_and_uncertainties`` -- ``MultimodalDEER`` (Stack C) or
    ``stackb.CompleteDEERModel`` (Stack B; its ``compute_loss`` is MultiTaskDEERLoss on its keys).  A model without a
    ``compute_loss`` gets ``val_loss`` = NaN.

    CCC / MAE / RMSE statistics are accumulated on the device batch by batch (mmdeer_eval_accumulate); only 24 doubles and
    the per-sample (mean error, mean uncertainty) pairs of the calibration error reach the host."""
    device = torch.device(device) if device is not None else next(model.parameters()).device
    model.eval()
    sm = StreamingMetrics(device)
    has_loss = hasattr(model, "compute_loss")
    losses, seen = [], False
    for loader in loaders.values():
        for batch in loader:
            a, v, t, y = unpack_batch(batch, device)
            out = model(a, v, t)
            p, u = model.get_predictions_and_uncertainties(out)
            if has_loss:
                losses.append(model.compute_loss(out, y)["total_loss"])
            sm.update(p, y.float(), u)
            seen = True
    if not seen:
        return {"val_loss": float("nan")}
    m = sm.compute()
    m["val_loss"] = float(torch.stack(losses).mean()) if losses else float("nan")
    return m


def evaluate_deer_model(model, test_loaders, device=None) -> Dict[str, float]:
    """evaluation.evaluate_deer_model (evaluation.py:785-808): metrics of the model on the test loaders (Stack C or B)."""
    m = evaluate_loaders(model, test_loaders, device)
    m["test_loss"] = m.pop("val_loss")
    return m


def profile_training_speed(model: MultimodalDEER, batch_size: int = 32, warmup: int = 10, iters: int = 100):
    """TrainingUtils.profile_training_speed (training.py:554-605): warm-up + timed forward and forward+backward."""
    from . import synth
    dev = next(model.parameters()).device
    b = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(batch_size).items()}
    out = {}
    for name, fn in (("forward", lambda: model(b["audio"], b["video"], b["text"])),
                     ("forward_backward", lambda: model.train_step(b["audio"], b["video"], b["text"], b["targets"]))):
        model.train(name != "forward")
        with torch.set_grad_enabled(False):
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        out[name] = {"ms_per_batch": dt * 1e3, "samples_per_sec": batch_size / dt}
    return out
