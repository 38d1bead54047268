"""Importable alias for the hyphenated package directory.

The product package lives in ``uncertainty-aware-multimodal-emotion-recognition_amd/``
(the name the build contract fixes); a hyphen cannot appear in an ``import``
statement, so this shim extends its own ``__path__`` with that directory and
``import mmdeer.model`` resolves to ``<that dir>/model.py``.
"""
import os as _os

_PKG_DIR = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "uncertainty-aware-multimodal-emotion-recognition_amd",
)
__path__.insert(0, _PKG_DIR)  # type: ignore[name-defined]
PACKAGE_DIR = _PKG_DIR

from ._version import __version__  # noqa: E402,F401
