"""squarna_amd -- MI355X-native folding core behind SQUARNA's Python API.

Drop-in for the single-sequence hot path of febos/SQUARNA:
``from squarna_amd import Predict, Main`` mirrors ``SQUARNA/__init__.py:1-2``.
"""
from .config import ParseConfig  # noqa: F401

__all__ = ["ParseConfig"]
