"""BASELINE config C4 shape on one GPU: N = 1500, D = 8 (unit + 2 categorical + 5 numeric), P outputs x 6
candidate structures through GPSearch.kernel_sweep (reference options: maxiter 50000, SciPy tolerances)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from waveome_amd import kernels as K
from waveome_amd.model_search import GPSearch
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(9102 + 4)
N = 1500
X = pd.DataFrame({"unit": rng.integers(0, 100, N).astype(float), "c1": rng.integers(0, 5, N).astype(float),
                  "c2": rng.integers(0, 3, N).astype(float)})
for d in range(5):
    X[f"x{d}"] = rng.uniform(-3, 3, N)
a = rng.normal(size=(P, 5)); w = rng.uniform(0.5, 3, size=(P, 5))
Y = pd.DataFrame({f"m{p}": sum(a[p, d] * np.sin(w[p, d] * (X[f"x{d}"] - X[f"x{d}"].mean()) / X[f"x{d}"].std()) for d in range(5))
                  + 0.5 * (X["c1"] == p % 5) + 0.1 * rng.normal(size=N) for p in range(P)})
gps = GPSearch(X, Y, unit_col="unit", categorical_vars=["c1", "c2"])
num = gps.cont_idx
mk = {"se": K.SquaredExponential, "m12": K.Matern12, "m32": K.Matern32, "m52": K.Matern52,
      "per": lambda active_dims: K.Periodic(K.SquaredExponential(active_dims=active_dims)), "lin": K.Lin}
structures = {nm: K.Sum([K.Categorical(active_dims=[gps.unit_idx])] + [f(active_dims=[c]) for c in num]) for nm, f in mk.items()}
t0 = time.perf_counter()
tab = gps.kernel_sweep(structures, keep_models=False)
dt = time.perf_counter() - t0
print(f"C4 shape: {P} outputs x 6 structures = {len(tab)} fits in {dt:.1f} s -> {len(tab)/dt:.1f} fits/s; "
      f"mean iterations {tab['nit'].mean():.1f}, evaluations {tab['nfev'].mean():.1f}; status counts {tab['status'].value_counts().to_dict()}")
print(tab.groupby('structure')[['bic','nit']].mean())
print(f"table checksum {float(np.nansum(tab['loss'].to_numpy() * np.arange(1, len(tab) + 1))):.10f}")  # (tools/run_c4_ranks.py prints the same)
if os.environ.get("WV_C4_TABLE"):   # per-task evaluation counts, e.g. for tools/simulate_dealing.py
    tab.reset_index()[["output", "structure", "nit", "nfev"]].to_csv(os.environ["WV_C4_TABLE"], index=False)
