import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
if os.path.join(REPO, "tests") not in sys.path:  # tests/played_world.py (spawned workers import test modules by name as well)
    sys.path.insert(0, os.path.join(REPO, "tests"))
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # kernel experiments: MAUA_TEST_LIB=tools/bin/libmaua_<name>.so (tools/build_exp.sh) runs the same suite against a variant
    # build of the library; unset (the driver's runs), the in-tree csrc/libmaua_hip.so is what loads
    alt = os.environ.get("MAUA_TEST_LIB")
    if alt:
        from maua_stylegan2_amd import _lib

        _lib.LIB_PATH = os.path.abspath(alt)
        print(f"[conftest] MAUA_TEST_LIB -> {_lib.LIB_PATH}")


@pytest.fixture(scope="session")
def built_lib():
    """libmaua_hip.so, (re)built in-tree when stale; hipcc cross-compiles without a GPU."""
    from maua_stylegan2_amd import build

    return build.build(verbose=False)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def gpu(built_lib):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from maua_stylegan2_amd import _lib

    _lib.load()
    return torch.device("cuda:0")
