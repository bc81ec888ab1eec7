"""The C4 sweep (tools/run_c4.py) under torch.distributed with several ranks - on a one-GPU box all of them on GPU 0 over
gloo - to rehearse the on-demand dealing at job size: every rank must end with the same full table, equal to a
single-process run when WV_C4_REF points at its CSV.  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node R --master-addr 127.0.0.1 --master-port P tools/run_c4_ranks.py [P_outputs]
(timings are not representative: the ranks share one GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch.distributed as dist

from waveome_amd import kernels as K
from waveome_amd import parallel
from waveome_amd.model_search import GPSearch

dist.init_process_group(os.environ.get("WV_DIST_BACKEND", "gloo"))
rank, world = parallel.dist_info()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
tab = gps.kernel_sweep(structures, keep_models=False, max_batch=int(os.environ.get("WV_C4_BATCH", "128")))
dt = time.perf_counter() - t0
st = gps.rank_stats
sig = float(np.nansum(tab["loss"].to_numpy() * np.arange(1, len(tab) + 1)))
print(f"rank {rank}/{world}: {len(tab)} fits in {dt:.1f} s; drew {st['tasks_per_rank']} tasks, rank seconds "
      f"{[round(v, 1) for v in st['rank_seconds']]}, max/mean {st['rank_time_max_over_mean']:.3f}; table checksum {sig:.10f}", flush=True)
if rank == 0 and os.environ.get("WV_C4_OUT"):
    tab.reset_index().to_csv(os.environ["WV_C4_OUT"], index=False)
dist.barrier()
dist.destroy_process_group()
