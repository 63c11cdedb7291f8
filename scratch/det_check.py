import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import build_pair
DT = 1e-3
from drake_amd import ARR as A
def run(mode, n=5):
    _, g = build_pair(seed=11)
    if os.environ.get("DET"): g.set_deterministic(True)
    for _ in range(n):
        if mode == "substep":
            g.substep(DT, -1)
        else:
            g.rebuild_mapping(False); g.calc_fem_state_and_force(DT); g.particle_to_grid(DT); g.update_grid(-1); g.grid_to_particle(DT)
    return g.download(A.POSITIONS), g.download(A.VELOCITIES)
a = run("substep"); b = run("substep"); c = run("phases"); d = run("phases")
for name, (x, y) in {"substep vs substep": (a, b), "phases vs phases": (c, d), "substep vs phases": (a, c)}.items():
    print(name, "pos", np.abs(x[0] - y[0]).max(), "vel", np.abs(x[1] - y[1]).max(), "n differing vel", int((x[1] != y[1]).any(1).sum()))
for n in (1, 2, 3):
    a = run("substep", n); c = run("phases", n)
    print(n, "steps: substep vs phases vel", np.abs(a[1] - c[1]).max(), int((a[1] != c[1]).any(1).sum()))
