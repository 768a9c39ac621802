import hashlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden():
    return json.load(open(os.path.join(GOLDEN_DIR, "golden.json")))


GOLDEN = load_golden()


def golden_frames(name):
    import vp8_oracle as vo
    w, h, frames = vo.read_ivf(os.path.join(GOLDEN_DIR, name + ".ivf"))
    return w, h, frames


def sha256(b):
    return hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="session")
def gpu_ctx():
    import alfalfa_amd as aa
    return aa.Context(0)   # raises NoDevice on a box without a GPU: -m gpu tests must not pass silently
