"""`from encoders import AudioEncoder, VideoEncoder, TextEncoder` (run_multimodal_deer.py:77).  The raw-signal front ends
(librosa / cv2 / BERT) are out of scope; the pre-extracted-feature branch of EnhancedAudioEncoder (row a14) is real."""
from mmdeer.side import EnhancedAudioEncoder  # noqa: F401


def _out_of_scope(name):
    class _Stub:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"encoders.{name}: raw-signal feature extraction is outside the mmdeer hot path (SURVEY.md 2); "
                                      "feed pre-extracted (B, 84) / (B, 256) / (B, 768) feature blocks")
    _Stub.__name__ = _Stub.__qualname__ = name
    return _Stub


AudioEncoder, VideoEncoder, TextEncoder = (_out_of_scope(n) for n in ("AudioEncoder", "VideoEncoder", "TextEncoder"))
