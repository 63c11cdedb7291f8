"""Per-iteration ratio |engine - oracle32| / runmax|oracle32 - oracle64| of the 20-iteration backtracking history, several
engine runs against ONE pair of oracle runs (the scene of tests/test_contact_gpu.py::_history_scene)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_contact_gpu import _history_scene, CONTACT_PARAMS
stiffness, damping, DT = CONTACT_PARAMS["config3"]
iters = 20
np.set_printoptions(precision=2, linewidth=250, suppress=True)
runmax = lambda x: np.maximum.accumulate(np.abs(x))
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    o, o64, g = _history_scene()
    o.update_contact(DT, 1.0, stiffness, damping, exact_line_search=False, max_iters=iters)
    o64.update_contact(DT, 1.0, stiffness, damping, exact_line_search=False, max_iters=iters)
    g.update_contact(DT, 1.0, stiffness, damping, exact_line_search=False, max_newton_iterations=iters)
    L32, L64, Lg = o.contact_log, o64.contact_log, g.contact_log().astype(np.float64)
    for name, (co, ce) in dict(residual=(6, 0), E1=(2, 2)).items():
        a, b, c = Lg[:, ce], L32[:, co], L64[:, co]
        print(k, name, "ratio", np.abs(a - b) / (runmax(b - c) + 1e-5 * np.abs(b) / 5))
        print(k, name, "rel   ", np.abs(a - b) / np.abs(b), flush=True)
    g.destroy()
