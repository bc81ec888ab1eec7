"""Closed-form log marginal likelihoods for kernels that are (numerically) of low rank - TEST INFRASTRUCTURE, like
oracle/gp_oracle.py: imported only by tests/, by tests' by-hand scripts and by bench.py's cpu_baseline leg, never by
waveome_amd/.  Independent of gp_oracle.py: nothing here forms a Gram matrix or a Cholesky factor of one.

  low_rank_lml            log N(y | 0, U diag(d) U' + s I) from the k x k capacitance matrix (determinant lemma, Woodbury)
  se_features             SquaredExponential as its Taylor feature map (exact to rounding on bounded inputs)
  periodic_features       gpflow Periodic(SquaredExponential) as its cosine series with modified-Bessel weights
  low_rank_predict        posterior mean / variance of f at new rows for the same model (Bayesian linear regression in the features)
  headline_features       the (U, d) pair of the headline structure, shared feature count for training and new inputs
  headline_closed_form    BASELINE's headline structure sum_d (SE[d] + Periodic(SE)[d]) + Constant mean: rank ~650
  spec_closed_form_grad   ... with exact derivatives for every variance parameter, the noise and the mean
  spec_features / spec_closed_form   ANY structure built from SE, Periodic, Lin, Poly, Constant, Categorical factors (sums of products)
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


def periodic_features(x, variance, lengthscale, period, J=None):
    """The Periodic(SquaredExponential) kernel v exp(-sin^2(pi tau / p) / (2 l^2)) as a cosine series: with a = 1 / (4 l^2),
    exp(-a) exp(a cos phi) = e^-a I_0(a) + 2 sum_j e^-a I_j(a) cos(j phi), phi = 2 pi tau / p - the modified Bessel functions
    decay factorially in j, so 2 J + 1 features (1, cos j phi_x, sin j phi_x) reproduce the kernel to rounding: a low-rank
    kernel, whose LML `low_rank_lml` gives exactly.  Nothing in here evaluates a sine of a DIFFERENCE or an exponential of it."""
    from scipy.special import ive

    a = 1.0 / (4.0 * lengthscale ** 2)
    if J is None:   # as many harmonics as carry weight: e^-a I_j(a) falls factorially once j > a
        J = 1
        while ive(J, a) > 1e-22 * ive(0, a) and J < 4000:
            J += 1
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


# ---------------------------------------------------------------------------------------------------------------------
# any structure whose factors are low rank: sums of products of SE / Periodic (on bounded inputs), Lin, Poly, Constant,
# Categorical - e.g. everything waveome's full_kernel_build makes from kerns = [SE, Periodic(SE), Lin]
# ---------------------------------------------------------------------------------------------------------------------
def _factor_features(f, theta, X, ref):
    """(U, w) with k_f(x, x') = sum_j w_j U_j(x) U_j(x'); `ref`: the inputs that fix the SE feature counts."""
    t, dims, p = f["type"], list(f["dims"]), f["p"]
    n = X.shape[0]
    if t == "const":
        return np.ones((n, 1)), np.array([theta[p[0]]])
    if t == "lin":
        return X[:, dims], np.full(len(dims), theta[p[0]])
    if t == "cat":
        lab = np.rint(X[:, dims[0]])
        levels = np.unique(np.rint(ref[:, dims[0]]))
        return (lab[:, None] == levels[None, :]).astype(float), np.full(len(levels), theta[p[0]])
    if t == "se":
        U, w = np.ones((n, 1)), np.array([theta[p[0]]])
        for d in dims:   # an SE over several columns is the product of one-column SEs (the variance counted once)
            nd = se_features(ref[:, d], 1.0, theta[p[1]])[0].shape[1]
            Ud, _ = _se_features_n(X[:, d], 1.0, theta[p[1]], nd)
            U = (U[:, :, None] * Ud[:, None, :]).reshape(n, -1)
            w = np.repeat(w, nd)
        return U, w
    if t == "periodic":
        if len(dims) != 1:
            raise NotImplementedError("Periodic on several columns")
        return periodic_features(X[:, dims[0]], theta[p[0]], theta[p[1]], theta[p[2]])
    if t == "poly":
        if len(dims) != 1:
            raise NotImplementedError("Poly on several columns")
        from math import comb

        x, v, off, deg = X[:, dims[0]], theta[p[0]], theta[p[1]], int(f["degree"])
        U = np.stack([x ** k for k in range(deg + 1)], axis=1)       # (v x x' + o)^deg = sum_k C(deg, k) v^k o^(deg-k) (x x')^k
        return U, np.array([comb(deg, k) * v ** k * off ** (deg - k) for k in range(deg + 1)])
    if t == "empty":
        return np.zeros((n, 1)), np.zeros(1)
    raise NotImplementedError(f"no finite feature map for kernel type {t!r} (the Matern family: see tests/statespace_matern.py)")


def spec_features(spec, theta, X, like=None, max_rank=8000):
    """(U, d) of a whole structure in oracle/gp_oracle.py's spec format (waveome_amd.ModelSpec.to_dict()): a product term's
    features are the row-wise Kronecker product of its factors' features, its weights their Kronecker product."""
    X = np.asarray(X, float)
    ref = X if like is None else np.vstack([np.asarray(like, float), X])
    cols, ds, total = [], [], 0
    for term in spec["terms"]:
        parts = [_factor_features(f, theta, X, ref) for f in term]
        rank = int(np.prod([u.shape[1] for u, _ in parts]))
        total += rank
        if total > max_rank:   # (products of wide feature maps: the capacitance matrix would not fit - not a closed form any more)
            raise ValueError(f"structure of rank > {max_rank}: no practical closed form")
        U, w = np.ones((X.shape[0], 1)), np.ones(1)
        for Uf, wf in parts:
            U = (U[:, :, None] * Uf[:, None, :]).reshape(X.shape[0], -1)
            w = (w[:, None] * wf[None, :]).reshape(-1)
        cols.append(U)
        ds.append(w)
    return np.column_stack(cols), np.concatenate(ds)


def spec_closed_form(spec, theta, X, y, max_rank=8000):
    """Exact LML of a low-rank structure (no Gram matrix): see spec_features."""
    U, d = spec_features(spec, theta, X, max_rank=max_rank)
    c = theta[spec["mean"]] if spec.get("mean") is not None else 0.0
    return low_rank_lml(U, d, theta[spec["noise"]], np.asarray(y, float) - c)


def low_rank_lml_grad(U, d, s, y):
    """(LML, dLML/dd [k], dLML/ds, dLML/dmean) for y ~ N(mean 1, U diag(d) U' + s I) at mean = 0 (pass y - mean).
    With C = I + D G / s, G = U'U:  K_y^-1 U = U C^-1 / s, so  u_j' K_y^-1 u_j = (G C^-1)_jj / s  and  U' alpha = C^-T U'y / s."""
    N, k = len(y), U.shape[1]
    D = np.diag(np.asarray(d, float))
    G = U.T @ U
    C = np.eye(k) + D @ G / s
    Cinv = np.linalg.inv(C)
    Uy = U.T @ y
    Ua = Cinv.T @ Uy / s                                   # U' alpha
    alpha = (y - U @ (Cinv @ (D @ Uy)) / s) / s            # K_y^-1 y
    logdet = N * np.log(s) + np.linalg.slogdet(C)[1]
    lml = -0.5 * (y @ alpha) - 0.5 * logdet - 0.5 * N * np.log(2 * np.pi)
    quad_j = np.einsum("ij,ji->i", G, Cinv) / s            # u_j' K_y^-1 u_j
    g_d = 0.5 * Ua * Ua - 0.5 * quad_j
    tr_kinv = (N - np.trace(Cinv @ D @ G) / s) / s         # tr K_y^-1 = (N - tr(C^-1 D G) / s) / s
    g_s = 0.5 * (alpha @ alpha) - 0.5 * tr_kinv
    return lml, g_d, g_s, float(np.sum(alpha))


def spec_closed_form_grad(spec, theta, X, y, max_rank=8000):
    """(LML, gradient over ALL parameters with NaN where this form has no analytic derivative): exact derivatives with
    respect to every kernel VARIANCE parameter (each feature weight is a monomial in them: its exponent is read off by
    doubling the parameter), the noise variance and the Constant mean.  Lengthscales, periods and Poly offsets move the
    features themselves - those entries are NaN; check them by central differences of `spec_closed_form`."""
    theta = np.asarray(theta, float)
    U, d = spec_features(spec, theta, X, max_rank=max_rank)
    c = theta[spec["mean"]] if spec.get("mean") is not None else 0.0
    lml, g_d, g_s, g_c = low_rank_lml_grad(U, d, theta[spec["noise"]], np.asarray(y, float) - c)
    grad = np.full(len(theta), np.nan)
    grad[spec["noise"]] = g_s
    if spec.get("mean") is not None:
        grad[spec["mean"]] = g_c
    var_params = sorted({f["p"][0] for term in spec["terms"] for f in term if f["type"] != "empty"})
    offsets = {f["p"][1] for term in spec["terms"] for f in term if f["type"] == "poly"}
    for p in var_params:
        if p in offsets:
            continue
        t2 = theta.copy()
        t2[p] *= 2.0
        d2 = _spec_weights(spec, t2, X)
        with np.errstate(divide="ignore", invalid="ignore"):
            expo = np.where(d != 0, np.log2(d2 / d), 0.0)
        grad[p] = float(np.sum(g_d * expo * d / theta[p]))
    return lml, grad


def _spec_weights(spec, theta, X):
    """The weights of spec_features alone (the features do not depend on the variance parameters)."""
    return spec_features(spec, theta, X)[1]

