import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _reset_margin_tag():
    yield
    try:
        from tests import helpers
        helpers.tag_default_engine(False)
    except Exception:  # noqa: BLE001
        pass


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
    for ratio, what, rtol, rel_scale, rel_plain in MARGINS:
        w = worst.get(what)
        if w is None:
            worst[what] = [ratio, rtol, rel_scale, rel_plain]
        else:
            w[0] = max(w[0], ratio); w[1] = max(w[1], rtol); w[2] = max(w[2], rel_scale); w[3] = max(w[3], rel_plain)
    lines = ["parity margins, worst case per quantity: error/allowed | tolerance | error rel. to the test's scale | "
             "error rel. to plain max|ref|"]
    for what, (ratio, rtol, rs, rp) in sorted(worst.items(), key=lambda kv: -kv[1][0]):
        lines.append(f"  {ratio:6.3f}  {rtol:8.1e}  {rs:9.2e}  {rp:9.2e}  {what}")
    for ln in lines[:16]:
        terminalreporter.write_line(ln)
    # the full table goes to a file next to the GPU logs when that directory exists
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_margins.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
