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


def se_features(x, variance, lengthscale):
    """The SquaredExponential kernel as its Taylor feature map: exp(-(x - x')^2 / 2 l^2) = e^{-z^2/2} e^{-z'^2/2} sum_n (z z')^n / n!,
    z = x / l - features e^{-z^2/2} z^n / sqrt(n!), as many as it takes for (max z^2)^n / n! to fall below 1e-20 (55 for
    |x| <= 3 l, 312 for |x| <= 10 l).  Exact to rounding on a bounded domain; no difference of inputs is ever formed."""
    from scipy.special import gammaln

    z = np.asarray(x, float) / lengthscale
    zmax = max(float(np.max(z * z)), 1e-300)
    n = 1
    while not (n > zmax and n * np.log(zmax) - gammaln(n + 1) < np.log(1e-20)):
        n += 1
    ns = np.arange(n)
    with np.errstate(divide="ignore"):
        logphi = ns[None, :] * np.log(np.abs(z))[:, None] - 0.5 * gammaln(ns + 1)[None, :] - 0.5 * (z * z)[:, None]
    U = np.where((z[:, None] < 0) & (ns[None, :] % 2 == 1), -1.0, 1.0) * np.exp(logphi)
    U[:, 0] = np.exp(-0.5 * z * z)          # (0^0 = 1)
    return U, np.full(n, float(variance))


def headline_closed_form(theta, X, y):
    """The BASELINE headline structure sum_d (SE[d] + Periodic(SE)[d]) + noise, Constant mean (bench.headline_spec: theta =
    [SE variance, lengthscale] per column, then [Periodic variance, lengthscale, period] per column, noise, c) as ONE
    low-rank kernel: exact LML at any N from a capacitance matrix of a few hundred rows."""
    D = X.shape[1]
    cols, ds = [], []
    for d in range(D):
        U, w = se_features(X[:, d], theta[2 * d], theta[2 * d + 1])
        cols.append(U)
        ds.append(w)
    for d in range(D):
        o = 2 * D + 3 * d
        U, w = periodic_features(X[:, d], theta[o], theta[o + 1], theta[o + 2])
        cols.append(U)
        ds.append(w)
    return low_rank_lml(np.column_stack(cols), np.concatenate(ds), theta[5 * D], np.asarray(y, float) - theta[5 * D + 1])


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


def main_headline(N):
    """The headline structure itself, one fit, at a size far beyond the bench's: device LML against the closed form."""
    import bench

    saved = bench.N_HEAD
    bench.N_HEAD = N
    try:
        X, Y = bench.make_inputs(1)
    finally:
        bench.N_HEAD = saved
    ms = bench.headline_spec()
    ctx = Context(0)
    prob = Problem(ctx, X, Y.T, ms, max_batch=1)
    t0 = time.time()
    want = headline_closed_form(ms.theta(), X, Y[0])
    tc = time.time() - t0
    t0 = time.time()
    lml, st = prob.lml([0], ms.theta()[None, :])
    t1 = time.time() - t0
    t0 = time.time()
    f, g, st2 = prob.value_grad([0], ms.unconstrain()[None, :])
    t2 = time.time() - t0
    prob.close()
    ctx.close()
    rel = max(abs(lml[0] - want), abs(-f[0] - want)) / abs(want)
    print(f"headline structure, N = {N}: closed form {want:.6f} ({tc:.1f} s on the CPU), device value-only {lml[0]:.6f} ({t1:.2f} s), "
          f"value + gradient {-f[0]:.6f} ({t2:.2f} s = {N ** 3 / t2 / 1e12:.1f} TFLOP/s on N^3); worst rel {rel:.1e}, status {st[0]} {st2[0]}")
    print("# ok" if rel < 1e-9 else "# MISMATCH")
    return 0 if rel < 1e-9 else 1


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    if len(sys.argv) > 2 and sys.argv[2] == "headline":
        return main_headline(N)
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
