"""Per-stage time of ONE value+gradient evaluation for a few kernel structures and batch sizes (HIP events around
every launch): where a round goes in the launch-bound regime."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem
ctx = Context(0)
rng = np.random.default_rng(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
X = rng.normal(size=(N, 3))
Y = np.stack([np.sin(X[:, 0] * (1 + 0.1 * p)) + 0.2 * rng.normal(size=N) for p in range(8)], axis=1)
specs = {"iso SE(0,1,2)": K.SquaredExponential(active_dims=[0, 1, 2]),
         "SE[0]+SE[1]+SE[2]": K.Sum([K.SquaredExponential(active_dims=[d]) for d in range(3)]),
         "headline 6 terms": K.Sum([K.SquaredExponential(active_dims=[d]) for d in range(3)]
                                   + [K.Periodic(K.SquaredExponential(active_dims=[d])) for d in range(3)])}
for name, kern in specs.items():
    spec = ModelSpec(kern, 3, K.GaussianLikelihood(), K.ConstantMean())
    for B in (1, 8):
        prob = Problem(ctx, X, Y, spec, max_batch=B)
        U = np.tile(spec.unconstrain(), (B, 1)); cols = np.arange(B, dtype=np.int32)
        prob.value_grad(cols, U)
        t0 = time.perf_counter()
        for _ in range(10): prob.value_grad(cols, U)
        wall = (time.perf_counter() - t0) / 10
        prob.timing(True); prob.timing_reset(); prob.value_grad(cols, U); st = prob.stage_times(); prob.timing(False)
        print(f"N={N} {name:20s} B={B}: wall {wall*1e3:6.3f} ms/eval; stages(ms) " +
              " ".join(f"{k.split('_')[0]}:{v['ms']:.3f}" for k, v in st.items() if v['launches']), flush=True)
        prob.close()
