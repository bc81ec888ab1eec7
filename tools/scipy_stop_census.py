"""How SciPy's own L-BFGS-B ends on the bench columns (CPU only, no GPU needed): `maxiter=100, ftol=gtol=0` on the
oracle objective (oracle/gp_oracle.py) for the first K columns of bench.make_inputs(512), one column per worker
process with single-threaded BLAS.  Prints nit / nfev / status / message / final loss per column - to be read next
to the GPU fits' table of the same columns (tools/fit_trajectory_parity.py writes it): do SciPy's fits stop before
iteration 100 as often, and the same ways, as the device fits do?  

Run:  python tools/scipy_stop_census.py K WORKERS > profiles/r03_scipy_stop_census.txt"""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402
import scipy.optimize  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(col):
    import bench
    from oracle import gp_oracle as o

    X, Y = bench.make_inputs(512)
    spec = bench.headline_spec_dict()
    u0 = o.pack_u(spec, o.theta_init(spec))
    t0 = time.perf_counter()
    res = scipy.optimize.minimize(lambda u: o.objective(spec, u, X, Y[col]), u0, jac=True, method="L-BFGS-B",
                                  options=dict(maxiter=100, maxfun=100000, ftol=0.0, gtol=0.0, maxcor=10, maxls=20))
    return col, res.nit, res.nfev, res.status, str(res.message), float(res.fun), time.perf_counter() - t0


if __name__ == "__main__":
    import multiprocessing as mp

    K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    print(f"# SciPy {scipy.__version__} L-BFGS-B on the oracle objective, bench columns 0..{K - 1}, maxiter=100, ftol=gtol=0")
    print("# column  nit  nfev  status  final_loss           seconds  message", flush=True)
    early = 0
    with mp.get_context("spawn").Pool(W) as pool:
        for col, nit, nfev, st, msg, f, dt in pool.imap(one, range(K)):
            early += nit < 100
            print(f"{col:8d} {nit:4d} {nfev:5d} {st:7d}  {f:.10f}  {dt:7.0f}  {msg}", flush=True)
    print(f"# {early} of {K} fits stopped before iteration 100")
