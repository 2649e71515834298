import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def fake_rna():
    """Installs tests/fake_rna.py as the module `RNA` (ViennaRNA stand-in, see its header) for the product's
    engine.vienna_bpp AND as the oracle's source of base-pair probabilities; restores both afterwards."""
    from tests import fake_rna as F
    from oracle import sqrn_oracle as O
    old_mod, old_src = F.install(), O.BPP_SOURCE
    O.BPP_SOURCE = O.ViennaBPP
    try:
        yield F
    finally:
        O.BPP_SOURCE = old_src
        F.uninstall(old_mod)
