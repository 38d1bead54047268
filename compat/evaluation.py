"""`from evaluation import evaluate_deer_model` (run_multimodal_deer.py:79; src/evaluation/evaluation.py:785-808)."""
from mmdeer.trainer import evaluate_deer_model, evaluate_loaders  # noqa: F401
