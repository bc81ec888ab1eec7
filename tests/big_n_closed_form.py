"""By-hand check at sizes beyond the unit tests (N = 16384 is the largest there): a kernel whose exact LML has a closed
form at any N - Constant(v) + Categorical(w)[two levels] + noise s, zero mean - so no O(N^3) oracle is needed:
K_y = v 11' + w (1_A 1_A' + 1_B 1_B') + s I is s I plus a rank-2 matrix; log|K_y| and the quadratic form follow from the
2 x 2 capacitance matrix (matrix determinant lemma / Woodbury).  Value-only path, value + gradient path (gradient
against central differences of the closed form).     python tests/big_n_closed_form.py [N=32768]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from waveome_amd import ModelSpec, kernels as K  # noqa: E402
from waveome_amd.engine import Context, Problem  # noqa: E402


def low_rank_lml(U, d, s, y):
    """log N(y | 0, U diag(d) U' + s I) from the k x k capacitance matrix (k = columns of U), exact at any N."""
    N = len(y)
    D = np.diag(np.asarray(d, float))
    C = np.eye(U.shape[1]) + D @ (U.T @ U) / s
    logdet = N * np.log(s) + np.linalg.slogdet(C)[1]
    Uy = U.T @ y
    quad = (y @ y - Uy @ np.linalg.solve(C, D @ Uy) / s) / s
    return -0.5 * quad - 0.5 * logdet - 0.5 * N * np.log(2 * np.pi)


def closed_form_lin_poly(v_lin, v_poly, offset, v_const, s, x, y):
    """Lin(v_lin)[x] + Poly(v_poly, offset, degree 2)[x] + Constant(v_const): (v x x' + o)^2 = v^2 x^2 x'^2 + 2 v o x x' + o^2
    (waveome/kernels.py:19-32, 56-73), so K = U diag(d) U' with U = [x, x^2, x, 1, 1]."""
    U = np.stack([x, x * x, x, np.ones_like(x), np.ones_like(x)], axis=1)
    return low_rank_lml(U, [v_lin, v_poly ** 2, 2 * v_poly * offset, offset ** 2, v_const], s, y)


def periodic_features(x, variance, lengthscale, period, J=90):
    """The Periodic(SquaredExponential) kernel v exp(-sin^2(pi tau / p) / (2 l^2)) as a cosine series: with a = 1 / (4 l^2),
    exp(-a) exp(a cos phi) = e^-a I_0(a) + 2 sum_j e^-a I_j(a) cos(j phi), phi = 2 pi tau / p - the modified Bessel functions
    decay factorially in j, so 2 J + 1 features (1, cos j phi_x, sin j phi_x) reproduce the kernel to rounding: a low-rank
    kernel, whose LML `low_rank_lml` gives exactly.  Nothing in here evaluates a sine of a DIFFERENCE or an exponential of it."""
    from scipy.special import ive

    a = 1.0 / (4.0 * lengthscale ** 2)
    phi = 2.0 * np.pi * np.asarray(x, float) / period
    cols, d = [np.ones_like(phi)], [variance * ive(0, a)]
    for j in range(1, J + 1):
        wj = 2.0 * variance * ive(j, a)
        cols += [np.cos(j * phi), np.sin(j * phi)]
        d += [wj, wj]
    return np.stack(cols, axis=1), np.array(d)


def closed_form_periodic(variance, lengthscale, period, v_lin, noise, x, y, mean=0.0):
    """Periodic(SE)[x] + Lin[x] + noise, Constant mean."""
    U, d = periodic_features(x, variance, lengthscale, period)
    U = np.column_stack([U, x])
    return low_rank_lml(U, np.concatenate([d, [v_lin]]), noise, np.asarray(y, float) - mean)


def closed_form(v, w, s, lab, y):
    N = len(y)
    U = np.stack([np.ones(N), (lab == 0).astype(float), (lab == 1).astype(float)], axis=1)   # K = U diag(v, w, w) U'
    D = np.diag([v, w, w])
    G = U.T @ U
    C = np.eye(3) + D @ G / s                       # det(K_y) = s^N det(I + D U'U / s)
    logdet = N * np.log(s) + np.linalg.slogdet(C)[1]
    Uy = U.T @ y
    quad = (y @ y - Uy @ np.linalg.solve(C, D @ Uy) / s) / s
    return -0.5 * quad - 0.5 * logdet - 0.5 * N * np.log(2 * np.pi)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    rng = np.random.default_rng(1)
    lab = rng.integers(0, 2, N).astype(float)
    X = np.column_stack([rng.normal(size=N), lab])
    y = 0.7 + 0.5 * (lab == 1) + 0.3 * rng.normal(size=N)
    v, w, s = 0.8, 0.4, 0.09
    kern = K.Sum([K.Constant(variance=v), K.Categorical(active_dims=[1], variance=w)])
    ms = ModelSpec(kern, 2, K.GaussianLikelihood(variance=s), mean_function=None)
    ctx = Context(0)
    t0 = time.time()
    prob = Problem(ctx, X, y[:, None], ms, max_batch=1)
    print(f"N = {N}: problem created in {time.time() - t0:.1f} s, workspace {prob.workspace_bytes / 2**30:.1f} GiB", flush=True)
    th = ms.theta()
    ref = closed_form(v, w, s, lab, y)
    t0 = time.time()
    lml, st = prob.lml([0], th[None, :])
    print(f"  value-only path: LML {lml[0]:.6f} vs closed form {ref:.6f}  (rel {abs(lml[0] - ref) / abs(ref):.1e}), status {st[0]}, "
          f"{time.time() - t0:.2f} s", flush=True)
    t0 = time.time()
    f, g, st = prob.value_grad([0], ms.unconstrain()[None, :])
    dt = time.time() - t0
    print(f"  value + gradient: loss {f[0]:.6f} (rel {abs(-f[0] - ref) / abs(ref):.1e}), {dt:.2f} s = {N ** 3 / dt / 1e12:.1f} TFLOP/s on N^3", flush=True)
    # gradient in the unconstrained coordinates against central differences of the closed form
    u = ms.unconstrain()
    num = np.zeros_like(u)
    for k in range(len(u)):
        h = 1e-5
        up, um = u.copy(), u.copy()
        up[k] += h
        um[k] -= h
        tp, tm = ms.constrain(up), ms.constrain(um)
        num[k] = -(closed_form(tp[0], tp[1], tp[2], lab, y) - closed_form(tm[0], tm[1], tm[2], lab, y)) / (2 * h)
    print(f"  gradient {g[0]} vs central differences {num}: max rel {np.max(np.abs(g[0] - num) / np.maximum(1.0, np.abs(num))):.1e}")
    prob.close()
    ctx.close()
    ok = abs(lml[0] - ref) / abs(ref) < 1e-9 and abs(-f[0] - ref) / abs(ref) < 1e-9 and \
        np.max(np.abs(g[0] - num) / np.maximum(1.0, np.abs(num))) < 1e-5
    print("# ok" if ok else "# MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
