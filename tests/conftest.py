"""pytest wiring: the `gpu` marker, repo-root imports, shared fixtures."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    os.environ.setdefault("IBU_TRACE_SORT", "1")   # the library says on stderr which path a sort took (test_gpu_sort.py asserts on it)
    # built artefacts are not in git: a fresh checkout builds them once (hipcc cross-compiles without a GPU)
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "ibu_amd", "libibu_hip.so")):
        subprocess.check_call(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "ibu_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libibu_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libibu_oracle.so"])


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    return orc

