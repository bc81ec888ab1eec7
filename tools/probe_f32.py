"""float32 path probes: GEMM kernel rates by shape / structure mode and the C5 stage breakdown."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _probes_lib  # noqa: F401  (diagnostic build: tools/libwv_probes.so)
from waveome_amd import ModelSpec, kernels as K  # noqa: E402
from waveome_amd.engine import Context, Problem  # noqa: E402

ctx = Context(0, dtype="float32")
for (M, N, Kd, nb) in [(4096, 4096, 4096, 1), (8192, 8192, 8192, 1), (2048, 2048, 2048, 16), (1024, 1024, 1024, 64)]:
    for a_kc in (False, True):
        for mode, name in [(0, "full"), (1, "lauum"), (2, "syrk"), (3, "khi2"), (6, "khi1")]:
            if mode in (1, 2) and M != N:
                continue
            tf, ms = ctx.gemm_bench(M, N, Kd, a_kcontig=a_kc, b_kcontig=False, bn=128 + 1000 * mode, nbatch=nb, iters=3)
            print(f"f32 gemm M={M} N={N} K={Kd} nb={nb} a_kc={int(a_kc)} {name:6s}: {tf:7.1f} TFLOP/s  {ms:8.3f} ms", flush=True)

import time

N, D, P = 16384, 4, 8
rng = np.random.default_rng(9102 + 5)
X = rng.uniform(-3, 3, size=(N, D))
X = (X - X.mean(0)) / X.std(0, ddof=1)
Y = np.stack([np.sin((1.0 + 0.2 * p) * X[:, 0]) + 0.5 * np.sin(2.1 * X[:, 1]) + 0.1 * rng.normal(size=N)
              for p in range(P)], axis=1)
spec = ModelSpec(K.Sum([K.Matern32(active_dims=[d]) for d in range(D)]), D, K.GaussianLikelihood(), K.ConstantMean())
th = spec.theta()
th[spec.noise] = 0.3
for dtype in ("float32", "float64"):
    c = ctx if dtype == "float32" else Context(0)
    for B in (1, 8):
        u = np.tile(spec.unconstrain(th), (B, 1))
        cols = np.arange(B, dtype=np.int32)
        prob = Problem(c, X, Y, spec, max_batch=B)
        prob.value_grad(cols, u)
        t0 = time.perf_counter()
        f, g, st_ = prob.value_grad(cols, u)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        prob.lml(cols, np.tile(th, (B, 1)))
        dv = time.perf_counter() - t0
        print(f"C5 {dtype} batch {B}: value+grad {dt * 1e3:8.1f} ms = {B * N ** 3 / dt / 1e12:6.1f} TFLOP/s (N^3 per fit);"
              f" value-only {dv * 1e3:8.1f} ms", flush=True)
        prob.timing(True)
        prob.value_grad(cols, u)
        st = prob.stage_times()
        prob.timing(False)
        for k, v in st.items():
            if v["launches"]:
                rate = f"{v['flops'] / (v['ms'] * 1e-3) / 1e12:6.1f} TFLOP/s" if v["flops"] else (
                    f"{v['bytes'] / (v['ms'] * 1e-3) / 1e9:7.1f} GB/s" if v["bytes"] else "")
                print(f"   {k:22s} {v['launches']:5d} launches {v['ms']:9.3f} ms  {rate}")
        prob.close()
