"""`from metrics import DEERMetrics` (run_multimodal_deer.py:80; src/utils/metrics.py)."""
from mmdeer.metrics import DEERMetrics, StreamingMetrics, uncertainty_calibration_error, validation_metrics  # noqa: F401
