"""Closed-form log marginal likelihoods for kernels that are (numerically) of low rank - TEST INFRASTRUCTURE, like
oracle/gp_oracle.py: imported only by tests/, by tests' by-hand scripts and by bench.py's cpu_baseline leg, never by
waveome_amd/.  Independent of gp_oracle.py: nothing here forms a Gram matrix or a Cholesky factor of one.

  low_rank_lml            log N(y | 0, U diag(d) U' + s I) from the k x k capacitance matrix (determinant lemma, Woodbury)
  se_features             SquaredExponential as its Taylor feature map (exact to rounding on bounded inputs)
  periodic_features       gpflow Periodic(SquaredExponential) as its cosine series with modified-Bessel weights
  low_rank_predict        posterior mean / variance of f at new rows for the same model (Bayesian linear regression in the features)
  headline_features       the (U, d) pair of the headline structure, shared feature count for training and new inputs
  headline_closed_form    BASELINE's headline structure sum_d (SE[d] + Periodic(SE)[d]) + Constant mean: rank ~650
  closed_form_lin_poly    Lin + Poly(degree 2) + Constant (waveome/kernels.py:19-32, 56-73)
  closed_form_periodic    Periodic(SE) + Lin
  closed_form             Constant + two-level Categorical (rank two: usable at N = 98304)
"""
import numpy as np


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


def low_rank_predict(U, d, s, y, Ustar):
    """Posterior mean and variance of f at rows Ustar for y ~ N(0, U diag(d) U' + s I): the Bayesian linear model in the
    feature weights w ~ N(0, diag(d)) - posterior covariance (diag(1/d) + U'U / s)^-1 = D - D G (I + D G)^-1 D ... written
    with the capacitance matrix C = I + D U'U / s so that zero weights d are fine: S = C^-1 D, m = S U'y / s."""
    D = np.diag(np.asarray(d, float))
    C = np.eye(U.shape[1]) + D @ (U.T @ U) / s
    S = np.linalg.solve(C, D)                       # posterior covariance of the weights
    m = S @ (U.T @ y) / s
    return Ustar @ m, np.einsum("ij,jk,ik->i", Ustar, S, Ustar)


def headline_features(theta, X, like=None):
    """(U, d) of the headline structure at inputs X; `like`: the inputs whose range fixes how many SE features are kept
    (pass the training inputs when X are new points, so that both use the same feature set)."""
    D_ = X.shape[1]
    ref = X if like is None else np.vstack([like, X])
    cols, ds = [], []
    for dd in range(D_):
        n = se_features(ref[:, dd], theta[2 * dd], theta[2 * dd + 1])[0].shape[1]
        U, w = _se_features_n(X[:, dd], theta[2 * dd], theta[2 * dd + 1], n)
        cols.append(U)
        ds.append(w)
    for dd in range(D_):
        o = 2 * D_ + 3 * dd
        U, w = periodic_features(X[:, dd], theta[o], theta[o + 1], theta[o + 2])
        cols.append(U)
        ds.append(w)
    return np.column_stack(cols), np.concatenate(ds)


def _se_features_n(x, variance, lengthscale, n):
    from scipy.special import gammaln

    z = np.asarray(x, float) / lengthscale
    ns = np.arange(n)
    with np.errstate(divide="ignore"):
        logphi = ns[None, :] * np.log(np.abs(z))[:, None] - 0.5 * gammaln(ns + 1)[None, :] - 0.5 * (z * z)[:, None]
    U = np.where((z[:, None] < 0) & (ns[None, :] % 2 == 1), -1.0, 1.0) * np.exp(logphi)
    U[:, 0] = np.exp(-0.5 * z * z)
    return U, np.full(n, float(variance))


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
