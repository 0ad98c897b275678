import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hostsim")):
    if p not in sys.path:
        sys.path.insert(0, p)

VOCABS = ["cl100k_base", "o200k_base", "llama3", "deepseek_v3", "mistral_v3"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        from splintr_amd import _ffi
        return _ffi.lib().spl_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # On a GPU box the gpu tests must RUN (and fail loudly if the extension is broken); on a box
    # without a GPU they are skipped unless explicitly selected with -m gpu.
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json"), encoding="utf-8") as f:
        g = json.load(f)
    return {k: v for k, v in g.items() if not k.startswith("_")}


@pytest.fixture(scope="session")
def golden_all():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json"), encoding="utf-8") as f:
        return json.load(f)


_coracles = {}


@pytest.fixture(scope="session")
def coracle():
    from oracle.coracle import COracle

    def get(name):
        if name not in _coracles:
            _coracles[name] = COracle(name)
        return _coracles[name]
    return get
