"""Differential fuzz at the FIT level: the device L-BFGS-B (FitQueue, several fits per batch) against SciPy's L-BFGS-B
on the oracle objective (what waveome/model_classes.py:310-334 runs), same start, same options, on seeded random
kernel trees and sizes.  Reports how often iteration and evaluation counts are identical, how far the end losses are
apart, and every case where one side ends more than 1e-6 (relative) above the other.  Trajectories that run long in a
flat valley amplify the last bits of the objective, so identical counts are the rule, not a law;
an end point that is WORSE by more than rounding would be a finding.  Where a trial point is not finite (a value of
+inf: K_y not positive definite, or a NaN in the gradient at absurd parameters) the two sides differ BY DESIGN (DESIGN.md
section 7, deviation 3): the device line search backs off toward the last finite point, SciPy's interpolation does not
recover - so every end-loss gap is reported with whether SciPy's run met such a point.  Not collected by pytest; run by hand:

    python tests/fuzz_fits.py [first_seed=7000] [n_seeds=400] > profiles/rNN_fuzz_fits.txt"""
import os
import sys
import time

import numpy as np
import scipy.optimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import test_gpu_parity as T  # noqa: E402
from oracle import gp_oracle as o  # noqa: E402
from waveome_amd import ModelSpec, kernels as K  # noqa: E402
from waveome_amd.engine import Context, Problem, lbfgs_opts  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    maxiter = 300
    ctx = Context(0)
    same_counts = total = worse_dev = worse_ref = both_fail = met_nonfinite = unexplained = 0
    gaps = []
    t0 = time.time()
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        kern = T._random_tree(rng, int(rng.integers(0, 3)))
        ms = ModelSpec(kern, 4, K.GaussianLikelihood(), K.ConstantMean())
        if rng.random() < 0.5:
            T._with_priors(ms, scale=float(rng.choice([0.1, 1.0, 10.0])))
        if sum(len(t) for t in ms.terms) > 128 or len(ms.terms) > 64 or max(len(t) for t in ms.terms) > 4:
            continue
        N = int(rng.choice([24, 50, 64, 100, 129, 200]))
        X, Y = T._data(seed, N)
        P = Y.shape[1]
        prob = Problem(ctx, X, Y, ms, max_batch=P)
        d = ms.to_dict()
        u0 = ms.unconstrain()
        res = prob.fit(np.arange(P, dtype=np.int32), np.tile(u0, (P, 1)), lbfgs_opts(maxiter=maxiter))
        prob.close()
        for c in range(P):
            nonfinite = [0]

            def counted(u, c=c):   # the oracle objective; counts trial points where the value or a gradient entry is not finite
                f, g = o.objective(d, u, X, Y[:, c])
                nonfinite[0] += not (np.isfinite(f) and np.all(np.isfinite(g)))
                return f, g

            rs = scipy.optimize.minimize(counted, u0, jac=True, method="L-BFGS-B",
                                         options={"maxiter": maxiter, "maxfun": maxiter, "ftol": 2.220446049250313e-09,
                                                  "gtol": 1e-05, "maxcor": 10, "maxls": 20})
            ref = {"loss": float(rs.fun), "nit": int(rs.nit), "nfev": int(rs.nfev), "status": int(rs.status),
                   "message": str(rs.message)}
            total += 1
            met_nonfinite += nonfinite[0] > 0
            fd, fr = float(res["loss"][c]), float(ref["loss"])
            if not np.isfinite(fr) or not np.isfinite(fd):
                both_fail += (not np.isfinite(fr)) and (not np.isfinite(fd))
                print(f"seed {seed} col {c}: N={N} params={ms.n_trainable}: device loss {fd}, SciPy loss {fr} "
                      f"(device status {int(res['status'][c])}; SciPy: {ref['message']})", flush=True)
                continue
            same = int(res["nit"][c]) == ref["nit"] and int(res["nfev"][c]) == ref["nfev"]
            same_counts += same
            gap = (fd - fr) / max(1.0, abs(fr))
            gaps.append(abs(gap))
            tag = ""
            if gap > 1e-6:
                worse_dev += 1
                tag = "  <-- device ends higher"
            elif gap < -1e-6:
                worse_ref += 1
                tag = "  <-- SciPy ends higher"
            if tag and nonfinite[0] == 0:
                unexplained += 1
                tag += " AND SciPy never saw a non-finite trial point"
            elif tag:
                tag += f" (SciPy met {nonfinite[0]} non-finite trial points)"
            if tag or not same:
                print(f"seed {seed} col {c}: N={N} params={ms.n_trainable}: device nit/nfev {int(res['nit'][c])}/"
                      f"{int(res['nfev'][c])} status {int(res['status'][c])} loss {fd:.10f}; SciPy {ref['nit']}/{ref['nfev']} "
                      f"status {ref['status']} loss {fr:.10f}; gap {gap:+.2e}{tag}", flush=True)
    gaps = np.array(gaps)
    print(f"# {total} fits ({n} seeds, maxiter {maxiter}, SciPy default tolerances) in {time.time() - t0:.0f} s: identical "
          f"iteration AND evaluation counts {same_counts} ({100.0 * same_counts / max(1, total):.1f} %); end losses: median gap "
          f"{np.median(gaps):.1e}, 99th percentile {np.percentile(gaps, 99):.1e}; device ends > 1e-6 higher: {worse_dev}, "
          f"SciPy ends > 1e-6 higher: {worse_ref}, of these {worse_dev + worse_ref} cases {unexplained} without a non-finite trial point "
          f"in SciPy's run ({met_nonfinite} of all fits met one); both sides without a finite value: {both_fail}")
    ctx.close()


if __name__ == "__main__":
    main()
