"""What is left of the float32 factor's error after the first-order correction: E = Mt K_y Mt' - I for the STORED float32
inverse factor Mt (numpy, float64 arithmetic on the downloaded matrices): its Frobenius norm (1/2 |E|_F^2 = the
second-order log-det term), spectrum, and how well a few Rademacher probes estimate |E|_F^2.
  python3 tools/probe_f32_residual.py [N]"""
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
    for noise in (0.3, 0.01):
        spec = ModelSpec(kern, D, K.GaussianLikelihood(), K.ConstantMean())
        th = spec.theta(); th[spec.noise] = noise
        p64 = Problem(c64, X, Y, spec, max_batch=1); Ky = p64.gram(th); l64, _ = p64.lml([0], th[None, :]); p64.close()
        p32 = Problem(c32, X, Y, spec, max_batch=1)
        Mt = np.tril(p32.dbg_stage(0, th, 1)[:N, :N]); l32, _ = p32.lml([0], th[None, :]); p32.close()
        E = Mt @ Ky @ Mt.T - np.eye(N)
        E = 0.5 * (E + E.T)
        fro2 = float(np.sum(E * E))
        ev = np.linalg.eigvalsh(E)
        exact_ld = -2 * np.sum(np.log(np.diag(Mt))) + np.sum(np.log1p(ev))
        first = -2 * np.sum(np.log(np.diag(Mt))) + np.trace(E)
        second = first - 0.5 * fro2
        third = second + np.sum(ev ** 3) / 3
        ref = np.linalg.slogdet(Ky)[1]
        est = []
        for m in (1, 4, 16):
            V = rng.choice([-1.0, 1.0], size=(N, m)); EV = E @ V
            est.append(float(np.sum(EV * EV) / m))
        print(f"N={N} {nm:9s} noise={noise}: cond(K_y) {np.linalg.cond(Ky):.2e}; |E|_F^2/2 = {fro2 / 2:.3e}, tr E = {np.trace(E):.3e}, "
              f"|E|_2 = {np.max(np.abs(ev)):.2e}, eig range [{ev[0]:.2e}, {ev[-1]:.2e}], participation (sum l^2)^2/sum l^4 = {fro2 ** 2 / np.sum(ev ** 4):.1f}; "
              f"log-det errors: first order {first - ref:.3e}, second {second - ref:.3e}, third {third - ref:.3e}, log1p(eig) {exact_ld - ref:.1e}; "
              f"Hutchinson |E|_F^2 with 1/4/16 probes: {[round(e / fro2, 4) for e in est]}; device lml f32 - f64 = {l32[0] - l64[0]:.3e}", flush=True)
