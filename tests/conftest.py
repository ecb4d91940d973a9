import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger / longer variant of a comparison whose smaller variant is in the default set; runs with "
                                       "--runslow or HSAD_RUN_SLOW=1 (a separate gpurun exercises them: tools/jobs/gpu_suite_slow.sh)")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False, help="also run the tests marked slow")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow") or os.environ.get("HSAD_RUN_SLOW"):
        return
    skip = pytest.mark.skip(reason="slow variant: --runslow / HSAD_RUN_SLOW=1 (its smaller variant runs by default)")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def hsad_lib():
    import hanabi_sad_amd
    return hanabi_sad_amd.load_library()
