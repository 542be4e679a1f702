import importlib
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- before librtx.so is loaded: torch brings its own HIP runtime (same soname), and a process
#                that has already mapped the system one through librtx.so leaves torch without a visible GPU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("ray-tracer_amd")


@pytest.fixture(scope="session")
def scenes(pkg):
    return importlib.import_module("ray-tracer_amd.scenes")


@pytest.fixture(scope="session")
def lib(pkg):
    return pkg.load_library()


@pytest.fixture(scope="session")
def lib_tuning(pkg):
    """The tuning build of the library (same sources, -DRTX_TUNING): reads the measurement switches such as RTX_FINE."""
    return pkg.load_library(pkg.LIB_TUNING_PATH)


@pytest.fixture(scope="session")
def orc(pkg):
    pyoracle = importlib.import_module("oracle.pyoracle")
    return pyoracle.Oracle(pkg)


@pytest.fixture(scope="session")
def refcore():
    pyoracle = importlib.import_module("oracle.pyoracle")
    try:
        return pyoracle.RefCore()
    except FileNotFoundError:
        pytest.skip("oracle/_ref not built (reference tree absent)")


@pytest.fixture(scope="session")
def core(pkg, lib):
    c = pkg.Core(0, lib)
    yield c
    c.close()


def pixel_centres(w, h):
    ys, xs = np.mgrid[0:h, 0:w]
    return np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], 1).astype(np.float32)
