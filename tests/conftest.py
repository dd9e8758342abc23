import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session", autouse=True)
def reference_summation_order():
    """The stage-level entry points sum as f64 trees by default (ae_set_summation_order, round 6); the parity tests compare them bit for
    bit with the oracle, which adds in the reference's sequential f32 order: ask for that order, for the whole session.  (Embedder.embed and
    EntropyOptim choose by their CE mode whatever this says.)  No GPU needed: the call only sets a flag."""
    try:
        import annembed_amd as A
        A.set_summation_order(False)
    except Exception:  # the library is not built: the tests that need it say so themselves
        pass
    yield
