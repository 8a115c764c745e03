import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "missing_artefacts(what): a build-container artefact the test needs is absent (tests/test_gpu_dropin.py: fails on a GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def ref_vectors(golden_dir):
    import json
    return json.load(open(os.path.join(golden_dir, "reference_unit_vectors.json")))
