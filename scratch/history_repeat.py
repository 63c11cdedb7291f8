"""Runs tests/test_contact_gpu.py::test_twenty_newton_iterations_decision_by_decision several times in one process and
prints the worst margin of each quantity per run (is a failure of that test the build's or the scene's spread?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from tests.test_contact_gpu import test_twenty_newton_iterations_decision_by_decision as t  # noqa: E402
import io, contextlib
for exact in (False, True):
    for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        helpers.MARGINS.clear()
        ok = True
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                t(exact)
        except AssertionError as exc:
            ok = False
            print('   ', str(exc).splitlines()[0][:300])
            print('   ', [l for l in buf.getvalue().splitlines() if l.startswith('20 iterations')])
        for l in buf.getvalue().splitlines():
            if l.startswith('20 iterations'):
                print('   ', l[l.index('rms engine'):])
        print("exact" if exact else "backtracking", k, "ok" if ok else "FAILED", " ".join(f"{m[1].split(': ')[-1]}={m[0]:.2f}" for m in helpers.MARGINS), flush=True)
