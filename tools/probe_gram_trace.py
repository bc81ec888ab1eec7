"""Gram assembly and gradient-trace pass of the headline structure at the bench's regime (N = 2048, 512 fits in one
batch): ms per launch (HIP events around the launches, `Problem.timing`), for whichever library WAVEOME_AMD_LIB names -
run once per library on the same box to compare builds (e.g. tools/libwv_probes.so against libwv_probes_r3exp.so,
or WV_GRAM_WAVES / WV_TRACE_WAVES = 3 in the diagnostic build).
usage (GPU box): python3 tools/probe_gram_trace.py [fits]"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, D = 2048, 3
ctx = Context(0)
rng = np.random.default_rng(0)
X = rng.normal(size=(N, D))
Y = np.stack([np.sin(X[:, 0] * (1 + 0.01 * p)) + 0.2 * rng.normal(size=N) for p in range(B)], axis=1)
kern = K.Sum([K.SquaredExponential(active_dims=[d]) for d in range(D)]
             + [K.Periodic(K.SquaredExponential(active_dims=[d])) for d in range(D)])
spec = ModelSpec(kern, D, K.GaussianLikelihood(), K.ConstantMean())
prob = Problem(ctx, X, Y, spec, max_batch=B)
U = np.tile(spec.unconstrain(), (B, 1)) + 0.1 * rng.normal(size=(B, spec.n_trainable))
cols = np.arange(B, dtype=np.int32)
f, g, st = prob.value_grad(cols, U)
prob.timing(True, ["gram_assembly", "grad_trace"])
prob.timing_reset()
for _ in range(5):
    prob.value_grad(cols, U)
t = prob.stage_times()
prob.timing(False)
tag = os.environ.get("WAVEOME_AMD_LIB", "release") + " " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("WV_"))
ga, tr = t["gram_assembly"], t["grad_trace"]
print(f"{tag}: fits {B}  gram {ga['ms'] / ga['launches']:.3f} ms  trace {tr['ms'] / tr['launches']:.3f} ms  "
      f"sum {ga['ms'] / ga['launches'] + tr['ms'] / tr['launches']:.3f} ms   checksum f {float(np.sum(f)):.17g} g {float(np.sum(g)):.17g}")
prob.close()
