"""The Newton-with-bisection root finder of the exact line search, pinned on the reference's own
test vectors (multibody/contact_solvers/test/newton_with_bisection_test.cc:55-225, transcribed as
data in tests/golden/newton_with_bisection_cases.json).

Both implementations are run: the oracle's (oracle/mpm_oracle.c, the code orc_update_contact calls)
and the product's (drake_amd/csrc/mpm_rootfind.h through the C ABI; host code, needs no GPU).
These are the only reference-held vectors that touch the hot path."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sgn_sqrt(x):
    s = 1.0 if x >= 0 else -1.0
    r = math.sqrt(abs(x))
    return s * r, (0.5 / r if r > 0 else math.inf)


# (f, f') of every case, restated from newton_with_bisection_test.cc:55-225
FUNCTIONS = {
    "linear": lambda x: (1.5 * x + 3.0, 1.5),
    "quadratic": lambda x: ((x - 1.5) * (x + 2.0), 2.0 * x + 0.5),
    "arctan": lambda x: (math.atan(x), 1.0 / (1.0 + x * x)),
    "cubic_cycle": lambda x: (x * (x * x - 2) + 2, 3 * x * x - 2),
    "signed_sqrt": _sgn_sqrt,
    "x_minus_tan": lambda x: (x - math.tan(x), -math.tan(x) * math.tan(x)),
    "triple_root": lambda x: ((x - 1.5) ** 3, 3.0 * (x - 1.5) ** 2),
    "kink": lambda x: (x, 1.0) if x < 1.0 else (2.0 * x, 2.0),
    "zero_slope_guess": lambda x: (x * (x - 1.0), 2.0 * x - 1.0),
}

with open(os.path.join(ROOT, "tests", "golden", "newton_with_bisection_cases.json")) as _f:
    CASES = json.load(_f)["cases"]

ORC_FN = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double))


def run_oracle(case, flags=0, f32=False):
    from oracle import oracle as orc
    L = orc.lib()
    fn = FUNCTIONS[case["fn"]]
    trace = []

    def cb(x, f, df):
        v = fn(x)
        trace.append(x)
        f[0], df[0] = v

    ev = C.c_int(0)
    if f32:
        root = C.c_float(0)
        rc = L.orc_newton_bisect_f32(ORC_FN(cb), C.c_float(case["a"]), C.c_float(case["b"]), C.c_float(case["guess"]),
                                     C.c_float(case["x_tol32"]), C.c_float(case["f_tol32"]), C.c_int(200), C.c_int(flags),
                                     C.byref(root), C.byref(ev))
    else:
        root = C.c_double(0)
        rc = L.orc_newton_bisect_f64(ORC_FN(cb), C.c_double(case["a"]), C.c_double(case["b"]), C.c_double(case["guess"]),
                                     C.c_double(case["x_tolerance"]), C.c_double(case["f_tolerance"]),
                                     C.c_int(case["max_iterations"]), C.c_int(flags), C.byref(root), C.byref(ev))
    return rc, root.value, ev.value, trace[2:]   # the first two calls evaluate the bracket ends


def run_engine(case, flags=0, f32=False):
    from drake_amd import capi
    L = capi.load_library()
    ENG_FN = capi.ROOTFIND_FN
    fn = FUNCTIONS[case["fn"]]
    trace = []

    def cb(_user, x, f, df):
        v = fn(x)
        trace.append(x)
        f[0], df[0] = v

    ev = C.c_int(0)
    if f32:
        root = C.c_float(0)
        rc = L.mpm_newton_bisect_f32(ENG_FN(cb), None, case["a"], case["b"], case["guess"], case["x_tol32"],
                                     case["f_tol32"], 200, flags, C.byref(root), C.byref(ev))
    else:
        root = C.c_double(0)
        rc = L.mpm_newton_bisect_f64(ENG_FN(cb), None, case["a"], case["b"], case["guess"], case["x_tolerance"],
                                     case["f_tolerance"], case["max_iterations"], flags, C.byref(root), C.byref(ev))
    return rc, root.value, ev.value, trace[2:]


@pytest.mark.parametrize("impl", ["oracle", "engine"])
@pytest.mark.parametrize("k", range(len(CASES)))
def test_reference_root_finding_cases(impl, k):
    """The reference's acceptance rule (newton_with_bisection_test.cc:251-260), Drake semantics."""
    case = CASES[k]
    rc, x, evals, _ = (run_oracle if impl == "oracle" else run_engine)(case)
    assert rc == 0, (case["description"], "did not converge")
    fx = FUNCTIONS[case["fn"]](x)[0]
    assert abs(x - case["root"]) < case["x_tolerance"] or abs(fx) < case["f_tolerance"], (case, x, fx)
    if case["num_iterations"] is not None:
        assert evals == case["num_iterations"]


@pytest.mark.parametrize("k", range(len(CASES)))
def test_oracle_and_engine_take_the_same_path(k):
    """Bit-identical iterates (doubles, same arithmetic): evaluation points, count and root."""
    case = CASES[k]
    for flags in (0, 7):
        ro, re = run_oracle(case, flags), run_engine(case, flags)
        assert ro[0] == re[0] and ro[2] == re[2]
        assert ro[1] == re[1]
        assert ro[3] == re[3]


@pytest.mark.parametrize("k", range(len(CASES)))
def test_float_clone_on_the_reference_cases(k):
    """The float clone with the solver's flags (cuda_mpm_solver.cu:383-471) on the same functions:
    it must still land on the root to float accuracy, oracle and engine step for step."""
    case = dict(CASES[k])
    eps32 = float(np.finfo(np.float32).eps)
    case["x_tol32"], case["f_tol32"] = 5 * eps32, 5 * eps32
    flags = 7
    ro, re = run_oracle(case, flags, f32=True), run_engine(case, flags, f32=True)
    assert ro[1] == re[1] and ro[2] == re[2] and ro[3] == re[3]
    x = ro[1]
    fx = FUNCTIONS[case["fn"]](x)[0]
    # a float iterate cannot do better than a few ulp of the root's magnitude
    assert abs(x - case["root"]) < 64 * eps32 * max(1.0, abs(case["root"])) or abs(fx) < 64 * eps32, (case, x, fx)
    assert ro[0] in (0, 1)


def test_the_clone_returns_the_updated_root():
    """MPM_RF_STEP_LAST: on |f| < f_tol the clone steps once more and returns that point
    (cuda_mpm_solver.cu:437-468); Drake returns the evaluated point."""
    case = dict(fn="quadratic", a=-1.0, b=2.0, guess=1.0, x_tolerance=1e-300, f_tolerance=1e-6, max_iterations=100)
    _, x_drake, n_drake, tr = run_engine(case, 0)
    _, x_clone, n_clone, tr2 = run_engine(case, 4)
    assert n_drake == n_clone and tr == tr2
    assert x_drake == tr[-1]
    assert x_clone != tr[-1] and abs(x_clone - 1.5) < abs(x_drake - 1.5)
