"""A deterministic stand-in for ViennaRNA's Python module ``RNA`` -- TEST INFRASTRUCTURE ONLY.

The reference reaches ViennaRNA (third party, not vendored, absent from this image) at
SQRNdbnseq.py:341-364 for every ``bpp != 0`` paramset.  The probabilities themselves stay
parity-unpinned; what CAN be pinned is everything around them: which calls are
made with which arguments, the ``max == 0`` rescale retry (:355-364), the skip when the retry is still
all zero, and the application ``scoremat *= (bpp/max)**p`` / ``+= (bpp/max)**-p`` plus all five
algorithms downstream.  ``tests/golden/gen_bpp_golden.py`` installs this module as ``RNA`` and runs the
REAL reference on it; the GPU tests install the same module and run the product
(``engine.vienna_bpp`` does its own ``import RNA``), so both sides see identical "probabilities".

The fake is a pure function of what the caller handed over:
  * the sequence given to ``fold_compound`` (after the reference's N-substitution, :343-344),
  * the SHAPE reactivities and (m, b) given to ``sc_add_SHAPE_deigan`` (:346-347), if any,
  * whether ``exp_params_rescale`` was called (:357).
Values come from a splitmix64 stream seeded by sha256 of those inputs (no dependence on numpy's
generators).  Two length classes exercise the reference's zero branches:
  len % 11 == 3 : all zeros until ``exp_params_rescale`` has been called (retry succeeds)
  len % 11 == 7 : all zeros always (matrix stays as it is)
"""
import hashlib
import struct

import numpy as np

_PAIRS = {"GC", "CG", "AU", "UA", "GU", "UG"}
CALLS = []          # (method, args) log of the last fold_compound, for tests that check the call sequence


def _splitmix64(seed, count):
    x = (np.uint64(seed) + np.arange(1, count + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)      # uniform [0, 1)


class fold_compound:
    def __init__(self, sequence):
        self.sequence = sequence
        self.shape = None
        self.rescaled = False
        self.pf_calls = 0
        del CALLS[:]
        CALLS.append(("fold_compound", sequence))

    def sc_add_SHAPE_deigan(self, reactivities, m, b):
        self.shape = (tuple(float(x) for x in reactivities), float(m), float(b))
        CALLS.append(("sc_add_SHAPE_deigan", self.shape))

    def pf(self):
        self.pf_calls += 1
        CALLS.append(("pf",))
        return ("." * len(self.sequence), 0.0)

    def mfe(self):
        CALLS.append(("mfe",))
        return ("." * len(self.sequence), -0.25 * len(self.sequence))

    def exp_params_rescale(self, mfe):
        assert mfe == -0.25 * len(self.sequence), "exp_params_rescale must get mfe()'s energy (SQRNdbnseq.py:356-357)"
        self.rescaled = True
        CALLS.append(("exp_params_rescale", mfe))

    def bpp(self):
        """(n+1) x (n+1), 1-based, upper triangle -- the shape ViennaRNA returns (:349 drops row/column 0)."""
        assert self.pf_calls > 0, "bpp() before pf()"
        CALLS.append(("bpp",))
        seq, n = self.sequence, len(self.sequence)
        out = np.zeros((n + 1, n + 1))
        if n % 11 == 7 or (n % 11 == 3 and not self.rescaled):
            return out
        h = hashlib.sha256()
        h.update(seq.encode("latin-1", "replace"))
        if self.shape is not None:
            h.update(struct.pack("<%dd" % len(self.shape[0]), *self.shape[0]))
            h.update(struct.pack("<2d", self.shape[1], self.shape[2]))
        h.update(b"R" if self.rescaled else b"-")
        seed = int.from_bytes(h.digest()[:8], "little")
        u = _splitmix64(seed, n * n).reshape(n, n)
        for i in range(n):
            for j in range(i + 4, n):
                if seq[i] + seq[j] in _PAIRS:
                    v = u[i, j]
                    out[i + 1, j + 1] = v * v * v if u[j, i] < 0.6 else 0.0      # sparse, skewed like real probabilities
        return out


def install():
    """sys.modules['RNA'] = this module; returns whatever was installed before (or None)."""
    import sys
    old = sys.modules.get("RNA")
    sys.modules["RNA"] = sys.modules[__name__]
    return old


def uninstall(old=None):
    import sys
    if old is None:
        sys.modules.pop("RNA", None)
    else:
        sys.modules["RNA"] = old
