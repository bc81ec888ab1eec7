"""CPU-only differential fuzz of the host L-BFGS-B (wv_lbfgs_*, the reverse-communication optimiser every device fit
runs) against SciPy's L-BFGS-B on the same smooth test functions from the same random starts: identical iteration and
evaluation counts are expected wherever rounding is not amplified; end values must agree.  No GPU.
    python tests/fuzz_lbfgs_host.py [n=600]"""
import os
import sys

import numpy as np
import scipy.optimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import test_host as H  # noqa: E402
from waveome_amd import _lib  # noqa: E402

lib = _lib.load()


def make(rng):
    kind = rng.choice(["quadratic", "rosenbrock", "logcosh", "quartic", "styblinski"])
    n = int(rng.integers(2, 25))
    if kind == "quadratic":
        Q = np.linalg.qr(rng.normal(size=(n, n)))[0]
        ev = 10.0 ** rng.uniform(0, rng.choice([2, 5, 8]), n)
        A = (Q * ev) @ Q.T
        b = rng.normal(size=n)
        return kind, n, lambda x: (0.5 * x @ A @ x - b @ x, A @ x - b)
    if kind == "rosenbrock":
        return kind, n, lambda x: (scipy.optimize.rosen(x), scipy.optimize.rosen_der(x))
    if kind == "logcosh":
        A = rng.normal(size=(2 * n, n))
        y = rng.normal(size=2 * n)
        return kind, n, lambda x: (np.sum(np.logaddexp(A @ x - y, y - A @ x)), A.T @ np.tanh(A @ x - y))
    if kind == "quartic":
        c = rng.uniform(0.1, 3, n)
        return kind, n, lambda x: (np.sum(c * x ** 4 + 0.5 * x ** 2) + 0.1 * np.sum(x) ** 2,
                                   4 * c * x ** 3 + x + 0.2 * np.sum(x))
    return kind, n, lambda x: (0.5 * np.sum(x ** 4 - 16 * x ** 2 + 5 * x), 0.5 * (4 * x ** 3 - 32 * x + 5))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    rng = np.random.default_rng(123)
    same = 0
    worst = 0.0
    by_kind = {}
    for case in range(n_cases):
        kind, n, fun = make(rng)
        x0 = rng.normal(size=n) * rng.choice([0.5, 2.0])
        maxiter = int(rng.choice([30, 200, 2000]))
        r, _ = H._run_scipy(fun, x0, maxiter=maxiter, maxfun=15000)
        m = H._run_mine(lib, fun, x0, maxiter=maxiter, maxfun=15000)
        eq = (m["nit"], m["nfev"]) == (r.nit, r.nfev)
        same += eq
        gap = abs(m["f"] - r.fun) / max(1.0, abs(r.fun))
        worst = max(worst, gap)
        k = by_kind.setdefault(kind, [0, 0, 0.0])
        k[0] += 1
        k[1] += eq
        k[2] = max(k[2], gap)
        if gap > 1e-6:
            print(f"case {case} {kind} n={n} maxiter={maxiter}: host {m['nit']}/{m['nfev']} f {m['f']:.12g} (task {m['task']}); "
                  f"SciPy {r.nit}/{r.nfev} f {r.fun:.12g} ({r.message}); gap {gap:.1e}")
    print(f"# {n_cases} cases: identical iteration and evaluation counts in {same} ({100 * same / n_cases:.1f} %); "
          f"largest end-value gap {worst:.1e}")
    for kind, (tot, eq, gap) in sorted(by_kind.items()):
        print(f"#   {kind:11s} {tot:4d} cases, identical counts {eq:4d}, largest gap {gap:.1e}")


if __name__ == "__main__":
    main()
