"""tests/test_contact_gpu.py::test_config3_full_size_against_the_oracle N times in one process: the spread of the converged
solve's distance to the oracle's (two solves that stop at different points of a noise-limited tail)."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from tests.test_contact_gpu import test_config3_full_size_against_the_oracle as t  # noqa: E402
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    helpers.MARGINS.clear()
    buf = io.StringIO()
    ok = True
    try:
        with contextlib.redirect_stdout(buf):
            t()
    except AssertionError as exc:
        ok = False
        print("   ", str(exc).splitlines()[0][:200])
    line = [l for l in buf.getvalue().splitlines() if l.startswith("1m converged")]
    m = {x[1]: x for x in helpers.MARGINS}
    rel = m.get("1m contact vel (rms, relative)")
    print(k, "ok" if ok else "FAILED", "rms %.4f max %.4f" % (rel[3], rel[4]) if rel else "", line[0] if line else "", flush=True)
