"""Minimal float32 workload for rocprofv3: config C5 (N = 16384, sum_d Matern32[d]), 8 outputs in one batch,
one value+gradient evaluation after a warm-up."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem
dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
N, D, P = 16384, 4, 8
rng = np.random.default_rng(9102 + 5)
X = rng.uniform(-3, 3, size=(N, D)); X = (X - X.mean(0)) / X.std(0, ddof=1)
Y = np.stack([np.sin((1.0 + 0.2 * p) * X[:, 0]) + 0.1 * rng.normal(size=N) for p in range(P)], axis=1)
spec = ModelSpec(K.Sum([K.Matern32(active_dims=[d]) for d in range(D)]), D, K.GaussianLikelihood(), K.ConstantMean())
th = spec.theta(); th[spec.noise] = 0.3
u = np.tile(spec.unconstrain(th), (P, 1))
prob = Problem(Context(0, dtype=dtype), X, Y, spec, max_batch=P)
for _ in range(2):
    prob.value_grad(np.arange(P, dtype=np.int32), u)
prob.close()
