import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with gzip.open(os.path.join(GOLDEN_DIR, name + ".json.gz"), "rb") as fh:
        return json.loads(fh.read().decode())


def golden_names():
    """The scoring fixtures (eps_* hold plots for fixtures listed here, breakmaf_* the MAF pre-splitting cases)."""
    return sorted(f[:-8] for f in os.listdir(GOLDEN_DIR) if f.endswith(".json.gz") and not f.startswith(("eps_", "breakmaf_")))


def hss_key(h):
    return (-h["score"], h["strand"], h["frame"], h["start"], h["end"])


@pytest.fixture(scope="session")
def goldens():
    return {n: load_golden(n) for n in golden_names()}
