"""`from training import DEERTrainer, TrainingConfig` (run_multimodal_deer.py:74; src/training/training.py)."""
from mmdeer.trainer import DEERTrainer, TrainingConfig, evaluate_loaders, profile_training_speed  # noqa: F401
