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


@pytest.fixture(scope="session", params=["packed", "dense"])
def gpu_ctx(request):
    """The session's shared context, once per coefficient storage format of the device parser: every -m gpu test that takes it
    runs in both (packed is the product's default; dense is what host-parsed frames use and what reconstruction reads)."""
    import alfalfa_amd as aa
    ctx = aa.Context(0)    # raises NoDevice on a box without a GPU: -m gpu tests must not pass silently
    ctx.set_packed_coefficients(request.param == "packed")
    assert ctx.info()["packed_coefficients"] == (1 if request.param == "packed" else 0)
    return ctx
