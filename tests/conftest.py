import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "survey_known_answers.json")) as f:
        return json.load(f)


def pytest_terminal_summary(terminalreporter):
    """How close every oracle-vs-engine comparison came to its tolerance (worst first)."""
    try:
        from tests.helpers import MARGINS
    except Exception:
        return
    if not MARGINS:
        return
    worst = {}
    for ratio, what in MARGINS:
        worst[what] = max(worst.get(what, 0.0), ratio)
    terminalreporter.write_line("parity margins (max error / allowed):")
    for what, ratio in sorted(worst.items(), key=lambda kv: -kv[1])[:12]:
        terminalreporter.write_line(f"  {ratio:6.3f}  {what}")
