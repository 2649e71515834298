"""squarna_amd -- MI355X-native folding core behind SQUARNA's Python API.

Drop-in for the single-sequence hot path of febos/SQUARNA:
``from squarna_amd import Predict, Main`` mirrors ``SQUARNA/__init__.py:1-2``.
"""
from .config import ParseConfig  # noqa: F401
from .api import Predict, Main  # noqa: F401
from .core import (BPMatrix, AnnotateStems, OptimalStems, RunAlgo, Edmonds, Hungarian, Nussinov,  # noqa: F401
                   SQRNdbnseq, RunSQRNdbnseq, ScoreStruct, ReferenceScores)


def BuildRfam(*args, **kwargs):
    """SQUARNA-build-rfam (SQRNrfam.py:301-316) downloads Rfam covariance models; it is
    outside the accelerated path and not part of this build."""
    raise NotImplementedError("BuildRfam is out of scope of squarna_amd (see DESIGN.md)")


__all__ = ["Predict", "Main", "BuildRfam", "ParseConfig", "BPMatrix", "AnnotateStems", "OptimalStems", "RunAlgo",
           "Edmonds", "Hungarian", "Nussinov", "SQRNdbnseq", "RunSQRNdbnseq"]
