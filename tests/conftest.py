import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain ``pytest`` on a box without a GPU skips the ``gpu`` tests instead of failing inside ``require_gpu``."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Prints how many free-running decode decisions were compared bit-exactly and how many fell behind a knife edge (tests/_margin.py)."""
    import _margin

    for line in _margin.summary_lines():
        terminalreporter.write_line(line)
    out = os.environ.get("MI355_MARGIN_REPORT")
    if out and _margin.REPORT:
        with open(out, "w") as f:
            f.write("\n".join(_margin.summary_lines()) + "\n")
