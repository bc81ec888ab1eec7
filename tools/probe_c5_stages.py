"""Per-stage times of one value + gradient evaluation at the C5 shape (N = 16384, 8 fits) on float32 and float64 contexts."""
import sys
sys.path.insert(0, ".")
import numpy as np
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem
N, D, P = 16384, 4, 8
rng = np.random.default_rng(9102 + 5)
X = rng.uniform(-3, 3, size=(N, D)); X = (X - X.mean(0)) / X.std(0, ddof=1)
Y = np.stack([np.sin((1.0 + 0.2 * p) * X[:, 0]) + 0.5 * np.sin(2.1 * X[:, 1]) + 0.1 * rng.normal(size=N) for p in range(P)], axis=1)
spec = ModelSpec(K.Sum([K.Matern32(active_dims=[d]) for d in range(D)]), D, K.GaussianLikelihood(), K.ConstantMean())
th = spec.theta(); th[spec.noise] = 0.3
u = np.tile(spec.unconstrain(th), (P, 1)); cols = np.arange(P, dtype=np.int32)
for dt in ("float32", "float64"):
    ctx = Context(0, dtype=dt); prob = Problem(ctx, X, Y, spec, max_batch=P)
    prob.value_grad(cols, u)
    prob.timing(True); prob.timing_reset(); prob.value_grad(cols, u); st = prob.stage_times(); prob.timing(False)
    tot = sum(v["ms"] for v in st.values())
    print(f"{dt}: " + "; ".join(f"{k} {v['ms']:.1f} ms" + (f" ({v['flops'] / v['ms'] / 1e9:.0f} TF)" if v['flops'] > 0 and v['ms'] > 0 else "") for k, v in st.items() if v["launches"]) + f"; total {tot:.1f} ms", flush=True)
    prob.close(); ctx.close()
