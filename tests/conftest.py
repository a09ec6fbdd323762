import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A gpu-marked test on a machine without a GPU is skipped, never silently
    # passed through some other path: the product has no CPU fallback.
    import nano_vllm_amd

    if nano_vllm_amd.device_count() > 0:
        # a GPU run gets its tree from the build container, where oracle/_ref/ was built from the reference: the
        # engine drop-in tests then FAIL rather than skip when that library did not travel (NVK_EXPECT_REF=0 opts out)
        os.environ.setdefault("NVK_EXPECT_REF", "1")
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
