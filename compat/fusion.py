"""`from fusion import HierarchicalMultimodalFusion` (run_multimodal_deer.py:78; src/models/fusion.py)."""
from mmdeer.fusions import AdaptiveFusionGating, AttentionFusion, BilinearFusion  # noqa: F401
from mmdeer.model import HierarchicalMultimodalFusion, create_fusion_module  # noqa: F401
