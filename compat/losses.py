"""`from losses import DEERLoss` (run_multimodal_deer.py:81; src/utils/losses.py: loss variant 2 and its companions)."""
from mmdeer.losses import (CalibrationLoss, CombinedDEERLoss, DEERLoss, MultiTaskDEERLoss,  # noqa: F401
                           UncertaintyRegularizationLoss, create_deer_loss)
