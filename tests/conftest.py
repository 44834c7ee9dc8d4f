import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # The product library from exactly this tree, BEFORE any test module is imported: a module that loads it while it is stale
    # keeps the stale one for the whole session (capi.load() caches).  A failure here is reported by the lx_lib fixture.
    try:
        from lambda_amd import build

        build.build_product()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session")
def lx_lib():
    from lambda_amd import build, capi

    # the library must have been compiled from exactly this tree: rebuilt here when it was not (hipcc is in the image on
    # the GPU box as well), and its own report is checked after loading
    build.build_product()
    lib = capi.load()
    assert lib.lx_build_id().decode() == build.source_id(), "liblambda_ext.so does not match the source tree"
    return lib


@pytest.fixture(scope="session")
def handle(lx_lib):
    """A live lx_handle on cuda:0 -- GPU tests only. Fails (not skips) when the HIP path is unusable."""
    from lambda_amd import capi

    h = capi.Handle(0)
    yield h
    h.close()
