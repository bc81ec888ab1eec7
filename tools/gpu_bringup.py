"""First-contact script for the GPU box: hardware probes, then stage timings at the bench size.
Writes gpurun_out/bringup.json.  Usage: python tools/gpu_bringup.py [N] [B]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _probes_lib  # noqa: F401  (diagnostic build: tools/libwv_probes.so)
from waveome_amd import ModelSpec, kernels as K  # noqa: E402
from waveome_amd.engine import Context, Problem, lbfgs_opts  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
out = {}
ctx = Context(0)
for wps in (1, 2):
    tf, ms = ctx.mfma_f64_rate(20000, wps)
    out[f"mfma_f64_tflops_wps{wps}"] = tf
    print(f"mfma f64 16x16x4 back-to-back, {wps} wave/SIMD: {tf:.1f} TFLOP/s ({ms:.2f} ms)", flush=True)
out["copy_GBps"] = ctx.copy_rate(1 << 30, 10)
print("copy GB/s", out["copy_GBps"], flush=True)

rng = np.random.default_rng(9102 + 1)
D = 3
X = rng.uniform(-3, 3, size=(N, D))
X = (X - X.mean(0)) / X.std(0, ddof=1)
P = 8
Y = np.stack([np.sin((1 + p) * X[:, 0]) + 0.5 * np.sin(1.7 * X[:, 1]) + 0.1 * rng.normal(size=N) for p in range(P)], 1)
kern = K.Sum([K.SquaredExponential(active_dims=[d]) for d in range(D)]
             + [K.Periodic(K.SquaredExponential(active_dims=[d])) for d in range(D)])
ms = ModelSpec(kern, D, mean_function=K.ConstantMean())
prob = Problem(ctx, X, Y, ms, max_batch=B)
print("workspace GB", prob.workspace_bytes / 1e9, flush=True)
cols = np.arange(B, dtype=np.int32) % P
U = np.tile(ms.unconstrain(), (B, 1))
prob.value_grad(cols, U)  # warm
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    prob.value_grad(cols, U)
dt = (time.perf_counter() - t0) / reps
out["value_grad_ms_per_batch"] = dt * 1e3
out["evals_per_s"] = B / dt
out["tflops_fp64_overall"] = B * float(N) ** 3 / dt / 1e12
print(f"value+grad: {dt*1e3:.2f} ms per batch of {B} -> {B/dt:.1f} evals/s, {out['tflops_fp64_overall']:.2f} TFLOP/s", flush=True)
prob.timing(True)
prob.timing_reset()
prob.value_grad(cols, U)
st = prob.stage_times()
prob.timing(False)
out["stages"] = st
for k, v in st.items():
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 and v["flops"] > 0 else 0
    gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 and v["bytes"] > 0 else 0
    print(f"  {k:22s} {v['ms']:9.3f} ms {v['launches']:5d} launches {tf:7.2f} TF/s {gb:8.1f} GB/s", flush=True)
t0 = time.perf_counter()
res = prob.fit(cols, U, lbfgs_opts(maxiter=20, ftol=0.0, gtol=0.0))
dt = time.perf_counter() - t0
out["fit20_s"] = dt
out["fit20_nfev"] = res["nfev"].tolist()
print(f"fit 20 its x {B} tasks: {dt:.2f} s, nfev mean {res['nfev'].mean():.1f}, status {set(res['status'].tolist())}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/bringup.json", "w") as fh:
    json.dump(out, fh, indent=1)
