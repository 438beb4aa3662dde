"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` tests run in the build container (no GPU): oracle vs golden vectors, host logic
through an oracle-backed test double, C-ABI symbol checks, gloo sharding tests.
`-m gpu` tests are the parity tests proper: HIP path through the C ABI vs oracle / goldens.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import hm_oracle
    hm_oracle.build()
    return hm_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
