"""Fit-level parity at the headline size (reference: waveome/model_classes.py:310-334, the
SciPy L-BFGS-B call this build's FitQueue replaces).

Fits all `P` bench columns on the GPU with the bench's options (maxiter=100, ftol=gtol=0), picks one column per
final status the bench line reports (3 = ran all 100 iterations, 2 = relative-reduction test with ftol=0, i.e. the
objective stopped changing, 5 = line search could not improve), then runs SciPy's L-BFGS-B on the CPU oracle
objective (oracle/gp_oracle.py) from the same start with the same options on those columns and prints, side by side:
nit, nfev, status / message, final loss, and the iteration up to which the per-iteration losses agree to 1e-9
relative (GPU iterate k = the GPU fit re-run with maxiter=k: fits are deterministic).

A checker, not product code: this is one of the places allowed to import oracle/.

Run on the GPU box:  python tools/fit_trajectory_parity.py [P=512] > profiles/r03_fit_trajectory_parity.txt
"""
import os
import sys
import time

import numpy as np
import scipy.optimize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle import gp_oracle as o  # noqa: E402
from waveome_amd.engine import Context, Problem, lbfgs_opts  # noqa: E402


def scipy_fit(spec_d, X, y, u0, maxiter):
    losses = []

    def fun(u):
        f, g = o.objective(spec_d, u, X, y)
        fun.last[tuple(u)] = f
        return f, g

    fun.last = {}

    def cb(xk):
        losses.append(fun.last.get(tuple(xk), np.nan))

    res = scipy.optimize.minimize(fun, u0, jac=True, method="L-BFGS-B", callback=cb,
                                  options=dict(maxiter=maxiter, maxfun=100000, ftol=0.0, gtol=0.0, maxcor=10, maxls=20))
    return res, np.array(losses)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    per_status = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    X, Y = bench.make_inputs(P)
    spec = bench.headline_spec()
    spec_d = spec.to_dict()
    ctx = Context(0)
    prob = Problem(ctx, X, Y.T, spec, max_batch=min(P, 512))
    cols = np.arange(P, dtype=np.int32)
    u0 = spec.unconstrain()
    U0 = np.tile(u0, (P, 1))
    opts = lbfgs_opts(maxiter=100, maxfun=100000, ftol=0.0, gtol=0.0)
    res = prob.fit(cols, U0, opts)
    st, cnt = np.unique(res["status"], return_counts=True)
    print(f"# GPU: {P} bench columns, maxiter=100, ftol=gtol=0: status counts "
          f"{dict(zip(st.tolist(), cnt.tolist()))}, iterations min {res['nit'].min()} mean {res['nit'].mean():.2f}")
    table_out = os.environ.get("WV_TABLE_OUT")   # the device fits' end states, column by column (read next to
    if table_out:                                 # tools/scipy_stop_census.py's table of SciPy's end states)
        with open(table_out, "w") as fh:
            fh.write("# device fits of bench.make_inputs(%d), maxiter=100, ftol=gtol=0\n# column nit nfev status final_loss\n" % P)
            for c in range(P):
                fh.write(f"{c:8d} {int(res['nit'][c]):4d} {int(res['nfev'][c]):5d} {int(res['status'][c]):6d}  {res['loss'][c]:.10f}\n")
    if per_status == 0:
        prob.close()
        ctx.close()
        return
    picks = []
    for s in (3, 2, 5):
        idx = np.flatnonzero(res["status"] == s)
        # the earliest stop of that kind is the most telling one
        idx = idx[np.argsort(res["nit"][idx], kind="stable")][:per_status]
        picks += [(int(c), s) for c in idx]
    print("# column  who     nit  nfev  status  final_loss             message")
    for c, s in picks:
        t0 = time.perf_counter()
        ref, ref_losses = scipy_fit(spec_d, X, Y[c], u0, 100)
        dt = time.perf_counter() - t0
        # GPU iterates: maxiter = k for k = 1..nit (deterministic re-runs, all in one batched call)
        kmax = int(res["nit"][c])
        gl = np.empty(kmax)
        for k0 in range(0, kmax, 1):
            r = prob.fit([c], u0[None, :], lbfgs_opts(maxiter=k0 + 1, maxfun=100000, ftol=0.0, gtol=0.0))
            gl[k0] = r["loss"][0]
        n = min(len(gl), len(ref_losses))
        rel = np.abs(gl[:n] - ref_losses[:n]) / np.abs(ref_losses[:n])
        bad = np.flatnonzero(~(rel <= 1e-9))
        agree = int(bad[0]) if len(bad) else n
        f_gpu_at_ref, _, _ = prob.value_grad([c], ref.x[None, :])
        f_ora_at_gpu, _ = o.objective(spec_d, res["u"][c], X, Y[c])
        print(f"{c:8d}  gpu   {int(res['nit'][c]):5d} {int(res['nfev'][c]):5d}  {int(res['status'][c]):6d}  "
              f"{res['loss'][c]:.12f}")
        print(f"{c:8d}  scipy {ref.nit:5d} {ref.nfev:5d}  {ref.status:6d}  {ref.fun:.12f}  {ref.message}  "
              f"({dt:.0f} s on the CPU)")
        print(f"{c:8d}  per-iteration losses agree to 1e-9 rel for the first {agree} of {n} iterations; "
              f"max rel diff over the first 10: {rel[:10].max():.2e}, first 30: {rel[:30].max():.2e}; "
              f"final losses differ by {abs(res['loss'][c] - ref.fun) / abs(ref.fun):.2e} rel; "
              f"objectives at each other's end point: gpu(scipy end) - scipy = {f_gpu_at_ref[0] - ref.fun:+.2e}, "
              f"oracle(gpu end) - gpu = {f_ora_at_gpu - res['loss'][c]:+.2e}")
        sys.stdout.flush()
    prob.close()
    ctx.close()


if __name__ == "__main__":
    main()
