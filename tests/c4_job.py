"""BASELINE config C4 at job shape (N = 1500, D = 8 = unit + 2 categorical + 5 numeric, P outputs x
6 candidate structures, fits run to the reference's tolerances) - the data, the structures and the entry point the
processes of tests/test_gpu_c4.py share.  The workload's origin is the reference's
examples/iHMP/ihmp_waveome_hpc_run.py:148-176; what the sweep replaces is the Ray round of
waveome/model_search.py:434-500.

As a script (one rank of a job; RANK / WORLD_SIZE set => a gloo process group):
    python tests/c4_job.py OUT_PICKLE P_OUTPUTS [MAXITER]"""
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N, SEED = 1500, 9102 + 4
STRUCTURES = ("se", "m12", "m32", "m52", "per", "lin")


def make_frames(P):
    import pandas as pd

    rng = np.random.default_rng(SEED)
    X = pd.DataFrame({"unit": rng.integers(0, 100, N).astype(float), "c1": rng.integers(0, 5, N).astype(float),
                      "c2": rng.integers(0, 3, N).astype(float)})
    for d in range(5):
        X[f"x{d}"] = rng.uniform(-3, 3, N)
    a = rng.normal(size=(P, 5))
    w = rng.uniform(0.5, 3, size=(P, 5))
    Y = pd.DataFrame({f"m{p}": sum(a[p, d] * np.sin(w[p, d] * (X[f"x{d}"] - X[f"x{d}"].mean()) / X[f"x{d}"].std())
                                   for d in range(5)) + 0.5 * (X["c1"] == p % 5) + 0.1 * rng.normal(size=N)
                      for p in range(P)})
    return X, Y


def make_search(P, columns=None):
    """GPSearch on the C4 inputs (optionally on a subset of the P output columns) and its six candidate structures:
    unit Categorical + one base kernel per numeric column."""
    from waveome_amd import kernels as K
    from waveome_amd.model_search import GPSearch

    X, Y = make_frames(P)
    if columns is not None:
        Y = Y[[f"m{p}" for p in columns]]
    gps = GPSearch(X, Y, unit_col="unit", categorical_vars=["c1", "c2"])
    mk = {"se": K.SquaredExponential, "m12": K.Matern12, "m32": K.Matern32, "m52": K.Matern52,
          "per": lambda active_dims: K.Periodic(K.SquaredExponential(active_dims=active_dims)), "lin": K.Lin}
    structures = {nm: K.Sum([K.Categorical(active_dims=[gps.unit_idx])] + [mk[nm](active_dims=[c]) for c in gps.cont_idx])
                  for nm in STRUCTURES}
    return gps, structures


def checksum(tab):
    return float(np.nansum(tab["loss"].to_numpy() * np.arange(1, len(tab) + 1)))


def main():
    out, P = sys.argv[1], int(sys.argv[2])
    maxiter = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(os.environ.get("WV_DIST_BACKEND", "gloo"))
    gps, structures = make_search(P)
    t0 = time.perf_counter()
    tab = gps.kernel_sweep(structures, keep_models=False, num_opt_iter=maxiter)
    dt = time.perf_counter() - t0
    res = {"table": tab.reset_index(), "seconds": dt, "checksum": checksum(tab),
           "rank_stats": getattr(gps, "rank_stats", None), "rank": rank, "world": world}
    with open(f"{out}.{rank}" if world > 1 else out, "wb") as fh:
        pickle.dump(res, fh)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
