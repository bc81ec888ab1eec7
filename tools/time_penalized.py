"""Where GPSearch.penalized_optimization() spends its wall time: batched fits vs per-output post-fit steps."""
import cProfile, pstats, sys, time
from pathlib import Path
import numpy as np, pandas as pd
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from waveome_amd.model_search import GPSearch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(1)
X = pd.DataFrame({"t": rng.uniform(0, 10, N), "dose": rng.normal(size=N), "age": rng.normal(size=N),
                  "grp": rng.integers(0, 4, N).astype(float)})
Y = pd.DataFrame({f"y{p}": np.sin(X["t"] * (1 + 0.1 * p)) + 0.5 * X["dose"] * (p % 3) + 0.3 * rng.normal(size=N) for p in range(P)})
gps = GPSearch(X, Y, categorical_vars=["grp"], Y_transform="standardize")
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
gps.penalized_optimization(penalization_factor=10.0)
pr.disable()
print(f"N={N} P={P}: total {time.time() - t0:.1f} s")
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(18)
