import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    from __graft_entry__ import load_oracle
    mod = load_oracle()
    mod.build()
    return mod


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def c2_image(pkg):
    """The BASELINE configs[1] index (2e8 rows, SURVEY.md 8(d) recipe, seed 42) as a .col_pml
    image in host memory, generated once per session (3.6 GB)."""
    rows = int(os.environ.get("COLBWT_TEST_ROWS", "200000000"))
    return pkg.synth_index(rows, mean_len=8, split_permille=0, seed=42)
