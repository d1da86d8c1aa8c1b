import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle with its C part built (gcc, a second or two)."""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(REPO, 'oracle')])
    from oracle import boxops_oracle
    return boxops_oracle
