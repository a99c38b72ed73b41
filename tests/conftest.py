import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def tables():
    from uuo_mocap_amd.body_model import synthetic_smpl

    return synthetic_smpl(0)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def oracle_smpl(tables):
    from oracle.smpl_ref import SmplInferenceRef

    return SmplInferenceRef(tables)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle_c():
    """oracle/_build/knn_cpu.so is produced by __graft_entry__.build(); build it on demand for CPU runs."""
    so = os.path.join(ROOT, "oracle", "_build", "knn_cpu.so")
    if not os.path.isfile(so):
        import subprocess

        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                               os.path.join(ROOT, "oracle", "knn_cpu.c"), "-lm"])
