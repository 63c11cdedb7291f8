// Newton-Raphson with bisection fallback on a bracketed root: the routine the exact line search of
// UpdateContact runs on dE/dalpha (cuda_mpm_solver.cu:383-471), itself a float clone of Drake's
// DoNewtonWithBisectionFallback (multibody/contact_solvers/newton_with_bisection.cc:14-119,
// Bracket: newton_with_bisection.h:21-60).
//
// Written as a resumable state machine (start / next evaluation point / feed the evaluation), so
// the same code drives a host loop, a single device lane between two reduction kernels, and the
// C-ABI test entry that runs the reference's own root-finding cases
// (multibody/contact_solvers/test/newton_with_bisection_test.cc:55-225) through it.
//
// `flags` select the clone's three deviations from the original:
//   RF_SIGN3      the bracket update compares a three-way sign (cuda_mpm_solver.cu:367-369, :431),
//                 the original compares signbit;
//   RF_NO_ENDS    no early return when an end of the bracket already meets f_tol (Drake :28-32);
//   RF_STEP_LAST  when |f(root)| < f_tol the clone still takes the step and returns the UPDATED
//                 root (:437-468); the original returns the evaluated one.
// The clone has no iteration limit; `max_evals` bounds the loop (a bracket of two adjacent floats
// whose Newton step stays above x_tol would otherwise spin for ever).
#pragma once

#ifdef __HIPCC__
#define RF_FN __host__ __device__ inline
#else
#define RF_FN inline
#endif

namespace mpm {

constexpr int RF_SIGN3 = 1, RF_NO_ENDS = 2, RF_STEP_LAST = 4;

template <class T>
struct RootFinder {
    T x_lo, f_lo, x_hi, f_hi, root, mdx, mdx_prev, x_tol, f_tol;
    int evals, max_evals, flags;
    int status;   // 0 running (evaluate at `root`, then feed), 1 converged, 2 out of evaluations

    static RF_FN T absT(T v) { return v < T(0) ? -v : v; }
    static RF_FN bool neg(T v) { return __builtin_signbit(v) != 0; }

    RF_FN void start(T xl, T fl, T xh, T fh, T guess, T xtol, T ftol, int max_ev, int fl_flags) {
        x_lo = xl; f_lo = fl; x_hi = xh; f_hi = fh;
        x_tol = xtol; f_tol = ftol; max_evals = max_ev; flags = fl_flags;
        evals = 0; status = 0;
        root = guess;
        mdx = x_lo - x_hi;
        mdx_prev = mdx;
        if (!(flags & RF_NO_ENDS)) {
            if (absT(f_lo) < f_tol) { root = x_lo; status = 1; return; }
            if (absT(f_hi) < f_tol) { root = x_hi; status = 1; return; }
        }
        if (max_evals <= 0) status = 2;
    }

    // f, df: the function and its derivative at `root`.  Afterwards either status != 0 (root is the
    // answer) or `root` is the next point to evaluate.
    RF_FN void feed(T f, T df) {
        evals += 1;
        bool differ;
        if (flags & RF_SIGN3) differ = ((f > T(0)) - (f < T(0))) != ((f_hi > T(0)) - (f_hi < T(0)));
        else differ = neg(f) != neg(f_hi);
        if (differ) { x_lo = root; f_lo = f; } else { x_hi = root; f_hi = f; }
        bool done = false;
        if (absT(f) < f_tol) {
            if (!(flags & RF_STEP_LAST)) { status = 1; return; }
            done = true;
        }
        const bool slow = T(2) * absT(f) > absT(mdx_prev * df);
        mdx_prev = mdx;
        if (slow) {
            mdx = T(.5) * (x_lo - x_hi);
            root = x_lo - mdx;
        } else {
            mdx = f / df;
            const T x = root - mdx;
            if (x_lo <= x && x <= x_hi) {
                root = x;
            } else {
                mdx = T(.5) * (x_lo - x_hi);
                root = x_lo - mdx;
            }
        }
        if (absT(mdx) < x_tol) done = true;
        if (done) status = 1;
        else if (evals >= max_evals) status = 2;
    }
};

}  // namespace mpm
