import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The plan picks the register-footprint scatters (k_grid_rec on single-pass plans with polynomial w-planes, k_grid_blk on
    # the others) only for problems with >= 2048 work items per pass -- the benchmark sizes -- and the single-launch
    # diagonal-walk scatter for small ones, which is what most parity cases are.  The parity suite therefore forces the
    # benchmark's kernels ("rec": the record scatter where the plan admits it, k_grid_blk elsewhere);
    # test_scatter_forms_agree covers all three forms and the automatic choice.
    os.environ.setdefault("PFBHIP_SCATTER", "rec")


def _have_gpu():
    try:
        from pfb_imaging_amd import _lib

        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests fail loudly (not skip) on a GPU box whose library is missing; on a box without a
    # GPU they are deselected by `-m "not gpu"`, and skipped if someone runs them anyway.
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
