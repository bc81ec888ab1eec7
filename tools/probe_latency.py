"""Launch-bound regime: wall time per L-BFGS evaluation for ONE fit (and a few) at several N, with and without
hipGraph replay of the evaluation pipeline (WV_GRAPH=0 / 1 in the environment of the run)."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem, lbfgs_opts
ctx = Context(0)
rng = np.random.default_rng(0)
for N in (150, 500, 1000, 2048, 4096):
    X = rng.normal(size=(N, 3))
    Y = np.stack([np.sin(X[:, 0] * (1 + 0.1 * p)) + 0.2 * rng.normal(size=N) for p in range(32)], axis=1)
    spec = ModelSpec(K.Sum([K.SquaredExponential(active_dims=[0]), K.Matern32(active_dims=[1])]), 3,
                     K.GaussianLikelihood(), K.ConstantMean())
    for B in (1, 8, 32):
        prob = Problem(ctx, X, Y, spec, max_batch=B)
        U0 = np.tile(spec.unconstrain(), (B, 1))
        cols = np.arange(B, dtype=np.int32)
        opts = lbfgs_opts(maxiter=30, ftol=0.0, gtol=0.0)
        prob.fit(cols, U0, opts)
        t0 = time.perf_counter()
        r = prob.fit(cols, U0, opts)
        dt = time.perf_counter() - t0
        rounds = int(r["nfev"].max())
        print(f"WV_BN_FILL={os.environ.get('WV_BN_FILL', 'default')} N={N:5d} fits={B}: {dt * 1e3:8.1f} ms for {rounds} rounds"
              f" = {dt * 1e3 / rounds:6.3f} ms per round; loss[0] {r['loss'][0]:.12g}", flush=True)
        prob.close()
