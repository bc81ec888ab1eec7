"""Job-size rehearsal of the sharded GPSearch paths: run the same job in ONE process and under torch.distributed with
several gloo ranks on the one GPU; the printed checksums must be identical (a fit depends only on its task).
  python tools/rehearse_ranks.py pen|pen_iter|pen_restart|sweep|search        (single process)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node R --master-addr 127.0.0.1 --master-port P tools/rehearse_ranks.py pen|search
pen:    GPSearch.penalized_optimization() - N = 900, 4 numeric + 1 categorical inputs, 24 outputs, kerns [SE, Periodic(SE)]
        (pen_iter: penalization_factor=None, three factor iterations; pen_restart: num_restart=3)
sweep:  GPSearch.kernel_sweep() on the same data: 3 structures x 24 outputs x 2 restarts
search: GPSearch.run_search(max_depth=3)  - N = 500, 3 inputs, 32 outputs, default kernel list (tools/run_search_demo.py's job)"""
import hashlib
import os
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from waveome_amd import kernels as K  # noqa: E402
from waveome_amd import parallel  # noqa: E402
from waveome_amd.model_search import GPSearch  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "pen"
if "RANK" in os.environ:
    import torch.distributed as dist

    dist.init_process_group(os.environ.get("WV_DIST_BACKEND", "gloo"))
rank, world = parallel.dist_info()
t0 = time.perf_counter()
if mode in ("pen", "pen_iter", "pen_restart", "sweep"):
    N, P = 900, 24
    rng = np.random.default_rng(9102 + 3)
    X = pd.DataFrame({f"x{d}": rng.uniform(-3, 3, N) for d in range(4)})
    X["cat"] = rng.integers(0, 5, N).astype(float)
    a = rng.normal(size=(P, 4)); w = rng.uniform(0.5, 3, size=(P, 4))
    Y = pd.DataFrame({f"y{p}": sum(a[p, d] * np.sin(w[p, d] * X[f"x{d}"] / X[f"x{d}"].std()) for d in range(4))
                      + 1.0 * (X["cat"] == p % 5) + 0.1 * rng.normal(size=N) for p in range(P)})
    gps = GPSearch(X, Y, categorical_vars=["cat"], Y_transform="standardize")
    if mode == "sweep":     # kernel_sweep: 24 outputs x 3 structures x 2 restarts
        structures = {"se": K.Sum([K.SquaredExponential(active_dims=[d]) for d in range(4)]),
                      "per": K.Sum([K.Periodic(K.SquaredExponential(active_dims=[d])) for d in range(2)]
                                   + [K.Categorical(active_dims=[4])]),
                      "lin*m32": K.Product([K.Lin(active_dims=[0]), K.Matern32(active_dims=[1])])}
        tab = gps.kernel_sweep(structures, num_restart=2, random_seed=11, keep_models=False)
        blob = tab[["loss", "lml", "bic", "nit", "nfev", "status"]].to_numpy(float).tobytes()
        extra = f"{len(tab)} (output, structure) pairs, evaluations mean {tab['nfev'].mean():.1f}"
    else:
        more = {"pen": {}, "pen_iter": {"penalization_factor": None, "num_factor_iter": 3},
                "pen_restart": {"num_restart": 3, "random_seed": 5}}[mode]
        gps.penalized_optimization(kernel_options={"second_order_numeric": False, "categorical_numeric_interactions": True,
                                                   "kerns": [K.SquaredExponential(), K.Periodic(K.SquaredExponential())]},
                                   max_batch=int(os.environ.get("WV_REH_BATCH", "0")) or None, **more)
        rec = np.asarray(gps.fit_records)
        blob = rec[:, :5].tobytes() + "|".join(
            f"{f}:{gps.models[f].kernel_name}:{gps.models[f].penalization_factor!r}" for f in gps.out_names).encode()
        extra = f"evaluations mean {rec[:, 3].mean():.1f}"
else:
    N, P = 500, 32
    rng = np.random.default_rng(0)
    t = rng.uniform(0, 10, N); dose = rng.normal(size=N); grp = rng.integers(0, 4, N).astype(float)
    X = pd.DataFrame({"t": t, "dose": dose, "grp": grp})
    cols = {}
    for p in range(P):
        eff = rng.normal(size=4)
        cols[f"y{p}"] = [np.sin(t), 0.8 * dose, eff[grp.astype(int)], np.sin(t) + 0.5 * dose][p % 4] + 0.3 * rng.normal(size=N)
    gps = GPSearch(X, pd.DataFrame(cols), categorical_vars=["grp"], Y_transform="standardize")
    gps.run_search(max_depth=3, keep_only_best=False)
    parts = []
    for f in gps.out_names:
        info = gps.search_info[f]
        parts.append(f + "=" + info["best_model"] + ";" + ";".join(f"{n}:{m['bic']!r}:{m['fit']['loss']!r}" for n, m in sorted(info["models"].items())))
    blob = "\n".join(parts).encode()
    extra = f"candidate fits {gps.search_stats['fits']}, structures {gps.search_stats['structures']}"
dt = time.perf_counter() - t0
st = getattr(gps, "rank_stats", None) or {}
print(f"{mode}: rank {rank}/{world} {dt:.1f} s; {extra}; tasks per rank {st.get('tasks_per_rank')}, max/mean "
      f"{st.get('rank_time_max_over_mean')}; checksum {hashlib.sha256(blob).hexdigest()[:20]}", flush=True)
if world > 1 or "RANK" in os.environ:
    dist.barrier()
    dist.destroy_process_group()
