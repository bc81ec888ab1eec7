"""How much of the float32 LML's remaining error is the rounding of the float32 product P = Mt' Mt?  The first-order
correction uses sum_ij (at at' - P)_ij K_ij with the P the device stored; exact arithmetic would use Mt' Mt of the stored
Mt.  numpy float64 on the downloaded matrices.   python3 tools/probe_f32_lauum_rounding.py [N]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = 4
rng = np.random.default_rng(9102 + 5)
X = rng.uniform(-3, 3, size=(N, D)); X = (X - X.mean(0)) / X.std(0, ddof=1)
Y = (np.sin(X[:, 0]) + 0.5 * np.sin(2.1 * X[:, 1]) + 0.1 * rng.normal(size=N))[:, None]
c32, c64 = Context(0, dtype="float32"), Context(0)
for nm, kern in {"sum M32": K.Sum([K.Matern32(active_dims=[d]) for d in range(D)]),
                 "SE + Per": K.Sum([K.SquaredExponential(active_dims=[0]), K.Periodic(K.SquaredExponential(active_dims=[1])), K.SquaredExponential(active_dims=[2])])}.items():
    spec = ModelSpec(kern, D, K.GaussianLikelihood(), K.ConstantMean())
    th = spec.theta(); th[spec.noise] = 0.3
    p64 = Problem(c64, X, Y, spec, max_batch=1); Ky = p64.gram(th); l64, _ = p64.lml([0], th[None, :]); p64.close()
    p32 = Problem(c32, X, Y, spec, max_batch=1)
    Mt = np.tril(p32.dbg_stage(0, th, 1)[:N, :N])
    P = np.tril(p32.dbg_stage(0, th, 2)[:N, :N]); P = P + np.tril(P, -1).T
    l32, _ = p32.lml([0], th[None, :]); p32.close()
    Pex = Mt.T @ Mt
    d = P - Pex
    E = Mt @ Ky @ Mt.T - np.eye(N)
    r = Y[:, 0] - th[spec.mean] if hasattr(spec, "mean") else Y[:, 0]
    print(f"N={N} {nm}: device lml f32 - f64 = {l32[0] - l64[0]:+.3e}; sum delta_ij K_ij / 2 = {0.5 * np.sum(d * Ky):+.3e} "
          f"(this enters the LML as -1/2 of it... sign checked below), |E|_F^2/4 = {0.25 * np.sum(E * E):.3e}; "
          f"delta: mean {d.mean():+.2e}, rms {np.sqrt((d * d).mean()):.2e}, diagonal mean {np.diag(d).mean():+.2e} (P_ii ~ {np.diag(Pex).mean():.2f}), "
          f"relative rms on the diagonal {np.sqrt(np.mean((np.diag(d) / np.diag(Pex)) ** 2)):.2e}", flush=True)
