import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line(
        'markers', 'needs_reference: imports the reference from /root/reference '
        '(build container only; skipped where it is absent)')


def pytest_collection_modifyitems(config, items):
    from oracle import refshim
    have_ref = refshim.available()
    skip_ref = pytest.mark.skip(reason='reference tree not present')
    for item in items:
        if 'needs_reference' in item.keywords and not have_ref:
            item.add_marker(skip_ref)
