"""`from preprocessing import create_enhanced_dataloaders` (run_multimodal_deer.py:75): out of scope."""


def create_enhanced_dataloaders(*a, **k):
    raise NotImplementedError("preprocessing.create_enhanced_dataloaders: dataset parsing is outside the mmdeer hot path (SURVEY.md 2)")
