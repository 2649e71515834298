"""squarna_amd -- MI355X-native folding core behind SQUARNA's Python API.

Drop-in for the single-sequence hot path of febos/SQUARNA:
``from squarna_amd import Predict, Main`` mirrors ``SQUARNA/__init__.py:1-2``.
"""
import os as _os

# The fold keeps several HIP streams busy at once (greedy rounds, the three matching kernels, the size classes of the
# blossom kernel, a second lane, and one set of those per batch in flight).  The ROCm runtime multiplexes all streams
# of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue serialise: measured
# on MI355X, 8 SRtest150 batches in flight fold in 19.6 ms with 16 queues and 29.5 ms with 4 (profiles/README.md,
# r02f).  The runtime reads the variable when it initialises (first HIP call), so it is set here, at import, unless
# the user has chosen a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .config import ParseConfig  # noqa: F401,E402
from .api import Predict, Main  # noqa: F401,E402
from .core import (BPMatrix, AnnotateStems, OptimalStems, RunAlgo, Edmonds, Hungarian, Nussinov,  # noqa: F401,E402
                   SQRNdbnseq, RunSQRNdbnseq, ScoreStruct, ReferenceScores)


def BuildRfam(*args, **kwargs):
    """SQUARNA-build-rfam (SQRNrfam.py:301-316) downloads Rfam covariance models; it is
    outside the accelerated path and not part of this build."""
    raise NotImplementedError("BuildRfam is out of scope of squarna_amd (see DESIGN.md)")


__all__ = ["Predict", "Main", "BuildRfam", "ParseConfig", "BPMatrix", "AnnotateStems", "OptimalStems", "RunAlgo",
           "Edmonds", "Hungarian", "Nussinov", "SQRNdbnseq", "RunSQRNdbnseq"]
