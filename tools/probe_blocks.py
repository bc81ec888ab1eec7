"""Block-diagonal structures (every term carries Categorical[unit]): time of one value+gradient round with
the zero-block-skipping products (default) and with WV_BLOCKS=0 (dense), rows sorted by unit and shuffled.
usage: python3 tools/probe_blocks.py [N] [units] [fits]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
units = int(sys.argv[2]) if len(sys.argv) > 2 else 32
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rng = np.random.default_rng(0)
per = N // units
uid = np.repeat(np.arange(units), per).astype(float)
uid = np.concatenate([uid, np.full(N - len(uid), units - 1.0)])
t = rng.uniform(-2, 2, N)
z = rng.normal(size=N)
X = np.column_stack([t, z, uid])
Y = np.stack([np.sin(2 * t + 0.3 * uid) + 0.3 * z + 0.1 * rng.normal(size=N) for _ in range(8)], axis=1)
u = lambda: K.Categorical(active_dims=[2])
kern = K.Sum([u(), K.Product([u(), K.SquaredExponential(active_dims=[0])]), K.Product([u(), K.Lin(active_dims=[1])]),
              K.Product([u(), K.Periodic(K.SquaredExponential(active_dims=[0]))])])
spec = ModelSpec(kern, 3, K.GaussianLikelihood(), K.ConstantMean())
ctx = Context(0)
cols = np.arange(B, dtype=np.int32) % 8
U = np.tile(spec.unconstrain(), (B, 1)) + 0.1 * rng.normal(size=(B, spec.n_trainable))
perm = rng.permutation(N)
ref = None
for label, Xc, Yc in (("rows sorted by unit", X, Y), ("rows shuffled", X[perm], Y[perm])):
    for mode in ("1", "0"):
        os.environ["WV_BLOCKS"] = mode
        p = Problem(ctx, Xc, Yc, spec, max_batch=B)
        f, g, st = p.value_grad(cols, U)
        t0 = time.perf_counter()
        for _ in range(5):
            f, g, st = p.value_grad(cols, U)
        dt = (time.perf_counter() - t0) / 5
        p.timing(True); p.timing_reset(); p.value_grad(cols, U); stg = p.stage_times(); p.timing(False)
        gemm = sum(v["ms"] for k, v in stg.items() if any(s in k for s in ("chol", "trtri", "lauum")))
        if ref is None:
            ref = f.copy()
        print(f"N={N} units={units} fits={B} {label:20s} WV_BLOCKS={mode}: {dt * 1e3:8.2f} ms per round "
              f"(products {gemm:7.2f} ms, gram {stg['gram_assembly']['ms']:.2f}, trace {stg['grad_trace']['ms']:.2f}); "
              f"max |loss - first run| {np.max(np.abs(f - ref)):.2e}", flush=True)
        p.close()
