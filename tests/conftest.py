import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")
    # pytest-timeout registers this one itself; declared here too so that the suite runs without the plugin
    config.addinivalue_line("markers", "timeout(seconds): fail a multi-process test that waits forever")


def pytest_collection_modifyitems(config, items):
    """The multi-process GPU tests (several processes sharing the box's card, each paging PyTorch in) go last: a slow
    or stuck rendezvous there must not stand between the single-process parity tests and their verdict."""
    late = [it for it in items if it.get_closest_marker("gpu") and it.module.__name__.endswith("test_distributed")]
    if late:
        items[:] = [it for it in items if it not in late] + late


@pytest.fixture(scope="session")
def engine():
    """One MI355X context shared by the GPU parity tests (fails loudly without a GPU)."""
    from mind_the_gaps_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


@pytest.fixture(autouse=True)
def _stacks_if_a_test_hangs():
    """A test still running after five minutes writes every thread's Python stack to stderr (and goes on): a hang
    on the GPU box then says where it waits instead of only timing the run out."""
    import faulthandler
    faulthandler.dump_traceback_later(300, exit=False)
    yield
    faulthandler.cancel_dump_traceback_later()
