"""BASELINE configs[2] through the user-facing API: N = 4096, D = 5 (4 numeric + 1 categorical),
64 output columns, `full_kernel_build(kerns=[SE, Periodic(SE)])` = 17 additive terms / 41 kernel parameters,
GPSearch.penalized_optimization() with the reference's options (run to SciPy's tolerances).
usage: python3 tools/run_c3.py [outputs = 64] [data seed offset = 0]   (other seeds: how much of the wall time is the
optimiser's path - tools/ab_c3_seeds.sh)"""
import sys, time
from pathlib import Path
import numpy as np, pandas as pd
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from waveome_amd import kernels as K
from waveome_amd.model_search import GPSearch
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 4096
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(9102 + 3 + SEED)
X = pd.DataFrame({f"x{d}": rng.uniform(-3, 3, N) for d in range(4)})
X["cat"] = rng.integers(0, 5, N).astype(float)
a = rng.normal(size=(P, 4)); w = rng.uniform(0.5, 3, size=(P, 4))
Y = pd.DataFrame({f"y{p}": sum(a[p, d] * np.sin(w[p, d] * (X[f"x{d}"] - X[f"x{d}"].mean()) / X[f"x{d}"].std()) for d in range(4))
                  + 1.0 * (X["cat"] == p % 5) + 0.1 * rng.normal(size=N) for p in range(P)})
gps = GPSearch(X, Y, categorical_vars=["cat"], Y_transform="standardize")
t0 = time.perf_counter()
gps.penalized_optimization(kernel_options={"second_order_numeric": False, "categorical_numeric_interactions": True,
                                           "unit_numeric_interactions": False,
                                           "kerns": [K.SquaredExponential(), K.Periodic(K.SquaredExponential())]},
                           penalization_factor=1.0)
dt = time.perf_counter() - t0
rec = gps.fit_records
print(f"C3 (data seed +{SEED}): {P} outputs, N={N}, 17 terms: penalized_optimization() {dt:.1f} s wall; iterations mean {rec[:, 2].mean():.0f} "
      f"max {rec[:, 2].max():.0f}; evaluations mean {rec[:, 3].mean():.0f}; status {np.unique(rec[:, 4], return_counts=True)}")
for f in list(gps.models)[:4]:
    print(" ", f, gps.models[f].kernel_name)
