"""`from multi_dataset_framework import MultiDatasetDEERFramework` (run_multimodal_deer.py:72): dataset parsing is out of
scope.  The name exists so the script's single import block gets past it; using it raises."""


class MultiDatasetDEERFramework:
    def __init__(self, *a, **k):
        raise NotImplementedError("multi_dataset_framework: dataset parsing is outside the mmdeer hot path (SURVEY.md 2); "
                                  "use experiments/run_multimodal_deer.py's synthetic loaders or your own DataLoaders")
