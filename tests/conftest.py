import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(autouse=True)
def _restore_library_options():
    """A test that steers a launcher with tell_amd.hip.set_option leaves the defaults (or what TELL_<KEY> asked for at load
    time) behind it."""
    hip = sys.modules.get('tell_amd.hip')
    before = None
    if hip is not None and getattr(hip, '_lib', None) is not None:
        before = {k: hip.get_option(k) for k in hip.option_keys()}
    yield
    hip = sys.modules.get('tell_amd.hip')
    if hip is not None and getattr(hip, '_lib', None) is not None:
        want = before if before is not None else hip.loaded_options()
        for k, v in want.items():
            if hip.get_option(k) != v:
                hip.set_option(k, v)


@pytest.fixture(scope='session')
def golden():
    import seeded

    def load(name):
        return seeded.load_npz(os.path.join(GOLDEN, name + '.npz'))
    return load
