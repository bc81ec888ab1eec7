"""Error budget of the float32 context's corrected LML at the C5 shape (numpy float64 on the downloaded matrices; several
minutes of host time at N = 16384).   python3 tools/probe_f32_budget.py [N]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, scipy.linalg as sl
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = 4
rng = np.random.default_rng(9102 + 5)
X = rng.uniform(-3, 3, size=(N, D)); X = (X - X.mean(0)) / X.std(0, ddof=1)
Y = (np.sin(X[:, 0]) + 0.5 * np.sin(2.1 * X[:, 1]) + 0.1 * rng.normal(size=N))[:, None]
spec = ModelSpec(K.Sum([K.Matern32(active_dims=[d]) for d in range(D)]), D, K.GaussianLikelihood(), K.ConstantMean())
th = spec.theta(); th[spec.noise] = 0.3
t0 = time.time()
c64 = Context(0); p64 = Problem(c64, X, Y, spec, max_batch=1); Ky = p64.gram(th); l64, _ = p64.lml([0], th[None, :]); p64.close(); c64.close()
c32 = Context(0, dtype="float32"); p32 = Problem(c32, X, Y, spec, max_batch=1)
Mt = np.tril(p32.dbg_stage(0, th, 1)[:N, :N]); P = np.tril(p32.dbg_stage(0, th, 2)[:N, :N]); l32, _ = p32.lml([0], th[None, :]); p32.close()
print(f"N={N}: device lml f64 {l64[0]:.6f}, f32 corrected {l32[0]:.6f}, diff {l32[0] - l64[0]:+.4e}   [{time.time() - t0:.0f} s]", flush=True)
r = Y[:, 0] - 0.0
L = sl.cholesky(Ky, lower=True, overwrite_a=False, check_finite=False)
ld = 2 * np.sum(np.log(np.diag(L))); zz = sl.solve_triangular(L, r, lower=True, check_finite=False); quad = zz @ zz
lml_np = -0.5 * quad - 0.5 * ld - 0.5 * N * np.log(2 * np.pi)
print(f"numpy lml {lml_np:.6f} (device f64 - numpy {l64[0] - lml_np:+.2e}); log|K| {ld:.6f}, quad {quad:.6f}   [{time.time() - t0:.0f} s]", flush=True)
del L
z = Mt @ r; al = Mt.T @ z; q = z @ z
slm = np.sum(np.log(np.diag(Mt)))
Kal = Ky @ al
# tr(P K) with the device's P (lower stored; exact float64 diagonal as the trace pass uses it) and with the exact product
pd = np.sum(Mt * Mt, axis=0)
trPK_dev = 2 * np.sum(np.tril(P, -1) * np.tril(Ky, -1)) + np.sum(pd * np.diag(Ky))
Pex = Mt.T @ Mt
trPK_ex = np.sum(Pex * Ky)
print(f"tr(P K) - N: device P (float64 diagonal) {trPK_dev - N:+.6e}, exact Mt'Mt {trPK_ex - N:+.6e}, difference {trPK_dev - trPK_ex:+.3e}; "
      f"off-diagonal delta: mean {np.mean(np.tril(P - Pex, -1)[np.tril_indices(N, -1)][:2000000]):+.2e}   [{time.time() - t0:.0f} s]", flush=True)
first = -q + 0.5 * (al @ Kal - trPK_ex) + slm + 0.5 * N - 0.5 * N * np.log(2 * np.pi)
print(f"first-order value with exact P {first:.6f}: error {first - lml_np:+.4e};  with the device's P: {first - 0.5 * (trPK_dev - trPK_ex) - lml_np:+.4e}", flush=True)
Ez = Mt @ Kal - z
V = rng.choice([-1.0, 1.0], size=(N, 8)); EV = Mt @ (Ky @ (Mt.T @ V)) - V
fro2 = np.sum(EV * EV, axis=0)
E2z = Mt @ (Ky @ (Mt.T @ Ez)) - Ez
print(f"|E zt|^2 = {Ez @ Ez:.4e}, zt'E^3 zt = {Ez @ E2z:.3e}, |E^2 zt|^2 = {E2z @ E2z:.3e}; |E|_F^2 by 8 single probes: {np.round(fro2, 5).tolist()} mean {fro2.mean():.5f}", flush=True)
second = first + 0.25 * fro2.mean() - 0.5 * (Ez @ Ez) + 0.5 * (Ez @ E2z) - 0.5 * (E2z @ E2z)
print(f"+ orders 2-4 (8-probe mean): error {second - lml_np:+.4e}   [{time.time() - t0:.0f} s]", flush=True)
