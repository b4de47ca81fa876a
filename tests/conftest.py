import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when collected without a device.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


def pytest_runtest_teardown(item, nextitem):
    """DM_HAZARD=1 python -m pytest -m gpu: every test runs under the stream-hazard tracker (dynamask_amd/hazard.py); a
    report raised by a test's launches fails THAT test (the planted-hazard test of test_hazard_gpu.py resets the tracker
    itself)."""
    import os
    if os.environ.get('DM_HAZARD', '0') in ('', '0'):
        return
    from dynamask_amd import hazard
    reports = hazard.reports()
    hazard.reset()
    if reports:
        import pytest
        pytest.fail('stream hazards reported during this test:\n' + '\n'.join(reports), pytrace=False)
