"""float32 context against float64 on the same inputs (C5-like: sum_d Matern32[d], sigma^2 = 0.3; and an SE + Periodic
mix): relative error of the loss / LML through both value paths and of the gradient (largest component).
  python3 tools/probe_f32_parity.py [N ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem

sizes = [int(a) for a in sys.argv[1:]] or [1000, 2048, 4096, 8192, 16384]
c32, c64 = Context(0, dtype="float32"), Context(0)
for N in sizes:
    D, P = 4, 2
    rng = np.random.default_rng(9102 + 5)
    X = rng.uniform(-3, 3, size=(N, D)); X = (X - X.mean(0)) / X.std(0, ddof=1)
    Y = np.stack([np.sin((1.0 + 0.2 * p) * X[:, 0]) + 0.5 * np.sin(2.1 * X[:, 1]) + 0.1 * rng.normal(size=N) for p in range(P)], axis=1)
    kerns = {"sum M32": K.Sum([K.Matern32(active_dims=[d]) for d in range(D)]),
             "SE + Per": K.Sum([K.SquaredExponential(active_dims=[0]), K.Periodic(K.SquaredExponential(active_dims=[1])),
                                K.SquaredExponential(active_dims=[2])])}
    for nm, kern in kerns.items():
        for noise in (0.3, 0.01):
            spec = ModelSpec(kern, D, K.GaussianLikelihood(), K.ConstantMean())
            th = spec.theta(); th[spec.noise] = noise
            u = np.tile(spec.unconstrain(th), (P, 1)); cols = np.arange(P, dtype=np.int32)
            out = {}
            for tag, ctx in (("f32", c32), ("f64", c64)):
                prob = Problem(ctx, X, Y, spec, max_batch=P)
                f, g, st = prob.value_grad(cols, u)
                t0 = time.perf_counter(); f, g, st = prob.value_grad(cols, u); ctx.synchronize(); tvg = time.perf_counter() - t0
                l, st2 = prob.lml(cols, np.tile(th, (P, 1)))
                t0 = time.perf_counter(); l, st2 = prob.lml(cols, np.tile(th, (P, 1))); ctx.synchronize(); tv = time.perf_counter() - t0
                out[tag] = (f, g, l, tvg, tv)
                prob.close()
            f32, g32, l32, tvg32, tv32 = out["f32"]; f64, g64, l64, tvg64, tv64 = out["f64"]
            print(f"N={N:6d} {nm:9s} noise={noise:5.2f}: loss rel err {np.max(np.abs(f32 - f64) / np.abs(f64)):.2e}, lml rel err "
                  f"{np.max(np.abs(l32 - l64) / np.abs(l64)):.2e} (abs {np.max(np.abs(l32 - l64)):.2e} of {abs(l64[0]):.1f}), grad err / max |g| "
                  f"{np.max(np.abs(g32 - g64)) / np.max(np.abs(g64)):.2e};  value+grad {tvg32 * 1e3:.1f} ms f32 / {tvg64 * 1e3:.1f} ms f64 "
                  f"({P * N ** 3 / tvg32 / 1e12:.1f} / {P * N ** 3 / tvg64 / 1e12:.1f} TF), value {tv32 * 1e3:.1f} / {tv64 * 1e3:.1f} ms", flush=True)
