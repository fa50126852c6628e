import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")
if ORACLE_DIR not in sys.path:
    sys.path.insert(0, ORACLE_DIR)

EPS = 2.220446049250313e-16


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_product():
    """Make sure libheyoka_amd.so exists (built in-tree, travels to the GPU box with the snapshot) and matches the sources:
    the import of heyoka_amd refuses a library whose build id differs from the hash of heyoka_amd/csrc (a stale prebuilt
    .so), in which case it is rebuilt here."""
    lib = os.path.join(ROOT, "heyoka_amd", "libheyoka_amd.so")
    stale = False
    if os.path.exists(lib):
        try:
            import heyoka_amd  # noqa: F401
        except ImportError as e:
            if "other sources" not in str(e):
                raise
            stale = True
    if stale or not os.path.exists(lib):
        import subprocess

        for m in [k for k in sys.modules if k == "heyoka_amd" or k.startswith("heyoka_amd.")]:
            del sys.modules[m]
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "heyoka_amd", "csrc"), "-j8"])


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "doc_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def outer_ss_golden():
    with open(os.path.join(ROOT, "tests", "golden", "outer_ss_ic.json")) as f:
        return json.load(f)


def sig_close(a, b, digits=6):
    """True if a agrees with the printed value b to the number of significant digits it was printed with."""
    import numpy as np

    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    scale = np.maximum(np.abs(b), 1e-300)
    return bool(np.all(np.abs(a - b) <= 0.6 * 10.0 ** (1 - digits) * scale))
