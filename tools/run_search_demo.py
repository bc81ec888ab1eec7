"""run_search on a C4-like shape: N x D inputs, P outputs, default kernel list; prints pooled-fit statistics."""
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from waveome_amd.model_search import GPSearch  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
P = int(sys.argv[2]) if len(sys.argv) > 2 else 32
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(0)
t = rng.uniform(0, 10, N)
dose = rng.normal(size=N)
grp = rng.integers(0, 4, N).astype(float)
X = pd.DataFrame({"t": t, "dose": dose, "grp": grp})
cols = {}
for p in range(P):
    kind = p % 4
    eff = rng.normal(size=4)
    base = [np.sin(t), 0.8 * dose, eff[grp.astype(int)], np.sin(t) + 0.5 * dose][kind]
    cols[f"y{p}"] = base + 0.3 * rng.normal(size=N)
Y = pd.DataFrame(cols)
gps = GPSearch(X, Y, categorical_vars=["grp"], Y_transform="standardize")
t0 = time.time()
gps.run_search(max_depth=depth, verbose=True)
dt = time.time() - t0
st = gps.search_stats
print(f"N={N} P={P} depth<={depth}: {st['fits']} fits over {st['structures']} structure batches in {dt:.1f} s"
      f" = {st['fits'] / dt:.1f} fits/s")
for p in range(min(P, 8)):
    print(f"  y{p} (truth kind {p % 4}):", gps.search_info[f"y{p}"]["best_model"], gps.search_info[f"y{p}"]["var_exp"])
