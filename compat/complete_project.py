"""`from complete_project import CompleteDEERModel, ModelConfig` (run_multimodal_deer.py:73) -> the HIP-backed model.
Default: mmdeer.model.MultimodalDEER (the fusion + DEER path of BASELINE.json); MMDEER_STACK=b selects
mmdeer.stackb.CompleteDEERModel, the restatement of the reference's own complete_project.CompleteDEERModel."""
import os

if os.environ.get("MMDEER_STACK", "c").lower() == "b":
    from mmdeer.stackb import CompleteDEERModel, ModelConfig, create_complete_deer_model  # noqa: F401
else:
    from mmdeer.model import CompleteDEERModel, ModelConfig, MultimodalDEER, create_model  # noqa: F401
    create_complete_deer_model = create_model
