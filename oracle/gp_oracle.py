"""CPU oracle for the per-output GP marginal-likelihood fitting path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``waveome_amd``) never routes through this file
and fails loudly when the HIP library is missing.

What it restates (NumPy/SciPy float64, dense, O(N^3) via LAPACK):

* the exact-GP log marginal likelihood the reference writes out in
  ``waveome/model_types_DEPR.py:42-56`` (``K -> K + sigma^2 I -> cholesky ->
  multivariate_normal``) and instantiates as ``gpflow.models.GPR`` in
  ``waveome/model_fitting.py:150-155``;
* the custom kernels of ``waveome/kernels.py`` (``Lin`` :5-39, ``Poly`` :42-83,
  ``Categorical`` :86-124, ``Empty`` :127-142) and the GPflow 2.9.1 kernels the
  reference composes with them (``regularization.py:23,112,139,166,181``):
  SquaredExponential, Matern12/32/52, Periodic(SquaredExponential), Constant,
  Sum, Product;
* the objective wiring of ``waveome/model_classes.py:817-864``: training loss
  ``-(LML + sum log Horseshoe(variance; scale=1/pf))`` over *trainable* kernel
  variance parameters, prior on the constrained value, no Jacobian;
* GPflow's parameter transforms (Softplus for positive parameters,
  Softplus + Shift(1e-6) for the Gaussian likelihood variance, identity for the
  Constant mean) and the SciPy ``L-BFGS-B`` call of
  ``waveome/model_classes.py:299-342`` (``gpflow.optimizers.Scipy().minimize``).

Third-party arithmetic (GPflow 2.9.1, TensorFlow-Probability 0.20-0.23, SciPy)
is not vendored under /root/reference and is absent from this image, so the
formulas below restate the published algorithms.  Pinning status:

* PINNED by the reference's own stored notebook outputs (tests/golden/kat.json):
  SquaredExponential value+optimum (KAT-1, 13 s.f.), Matern52 optimum (KAT-2),
  Matern12 optimum to 2 dp (KAT-3), Softplus / Softplus+Shift transforms,
  the LML formula, GPflow default initial values.
* PARITY UNPINNED (no known answer anywhere in the reference): Matern32,
  Periodic, Categorical, Lin, Poly, Sum/Product composites, the Constant mean,
  and the Horseshoe log-density.  They are checked by finite differences and
  algebraic identities only (tests/test_oracle.py).

Kernel-spec format (plain data, shared with the product's encoder only as a
dict so the oracle imports nothing from the product):

    spec = {
      "D": int,
      "params": [{"name": str, "init": float, "transform": "softplus" |
                  "softplus_shift" | "identity", "trainable": bool,
                  "prior_scale": float | None}, ...],
      "terms": [[factor, ...], ...],          # K = sum_t prod_f k_f
      "noise": int,                           # index into params
      "mean": int | None,                     # Constant mean c
    }
    factor = {"type": "se"|"m12"|"m32"|"m52"|"periodic"|"lin"|"poly"|"const"|
                      "cat"|"empty",
              "dims": [int, ...], "p": [param indices ...], "degree": int}
    param order inside "p": se/m12/m32/m52 -> [variance, lengthscales];
    periodic -> [variance, lengthscales, period]; lin/const/cat -> [variance];
    poly -> [variance, offset]; empty -> [].
"""

from __future__ import annotations

import math

import numpy as np
import scipy.linalg
import scipy.optimize

NOISE_FLOOR = 1e-6  # GPflow Gaussian likelihood DEFAULT_VARIANCE_LOWER_BOUND [3P]
LOG_2PI = math.log(2.0 * math.pi)
SQRT3 = math.sqrt(3.0)
SQRT5 = math.sqrt(5.0)


# --------------------------------------------------------------------------
# transforms (GPflow `positive()` = Softplus; likelihood variance = Softplus+Shift)
# --------------------------------------------------------------------------
def softplus(u):
    u = np.asarray(u, dtype=np.float64)
    return np.logaddexp(0.0, u)


def inv_softplus(x):
    x = np.asarray(x, dtype=np.float64)
    # log(expm1(x)) written to stay finite for large x
    return x + np.log(-np.expm1(-x))


def sigmoid(u):
    """d softplus / du, evaluated without cancellation for u << 0 (a variance driven to 1e-15 by
    the Horseshoe prior sits at u ~ -34, where 0.5 (1 + tanh(u/2)) has ~5 % rounding error)."""
    u = np.asarray(u, dtype=np.float64)
    e = np.exp(-np.abs(u))
    return np.where(u >= 0, 1.0 / (1.0 + e), e / (1.0 + e))


U_FLOOR = -340.0  # positive parameters saturate at softplus(-340) ~ 2e-148 (never 0, and (x/s)^2 stays a
                  # normal double, so the Horseshoe log-density stays finite); below the floor
                  # d theta / du = 0.  The product mirrors this (wv_spec.cpp).


def constrain_one(kind, u):
    u = max(float(u), U_FLOOR) if kind != "identity" else u
    if kind == "softplus":
        return float(softplus(u))
    if kind == "softplus_shift":
        return float(NOISE_FLOOR + softplus(u))
    if kind == "identity":
        return float(u)
    raise ValueError(kind)


def unconstrain_one(kind, x):
    if kind == "softplus":
        return float(inv_softplus(x))
    if kind == "softplus_shift":
        return float(inv_softplus(x - NOISE_FLOOR))
    if kind == "identity":
        return float(x)
    raise ValueError(kind)


def dconstrain_one(kind, u):
    if kind in ("softplus", "softplus_shift"):
        return float(sigmoid(u)) if u >= U_FLOOR else 0.0
    return 1.0


def trainable_index(spec):
    return [i for i, p in enumerate(spec["params"]) if p["trainable"]]


def theta_init(spec):
    return np.array([p["init"] for p in spec["params"]], dtype=np.float64)


def pack_u(spec, theta):
    """constrained full vector -> unconstrained trainable vector."""
    return np.array(
        [unconstrain_one(spec["params"][i]["transform"], theta[i]) for i in trainable_index(spec)],
        dtype=np.float64,
    )


def unpack_u(spec, u, theta_base=None):
    """unconstrained trainable vector -> constrained full vector."""
    theta = theta_init(spec) if theta_base is None else np.array(theta_base, dtype=np.float64)
    for k, i in enumerate(trainable_index(spec)):
        theta[i] = constrain_one(spec["params"][i]["transform"], u[k])
    return theta


# --------------------------------------------------------------------------
# base kernels: value and d/d(param) on dense (N, N2) grids
# --------------------------------------------------------------------------
def _sqdist(X, X2, dims):
    r2 = np.zeros((X.shape[0], X2.shape[0]))
    for d in dims:
        diff = X[:, d][:, None] - X2[:, d][None, :]
        r2 += diff * diff
    return r2


def factor_value_and_grads(f, theta, X, X2=None):
    """Return (k, [dk/dparam for param in f['p']]) as dense arrays.

    Formulas: SURVEY.md Appendix A; custom kernels waveome/kernels.py:19-32
    (Lin), :56-73 (Poly), :95-117 (Categorical), :136-139 (Empty)."""
    if X2 is None:
        X2 = X
    t = f["type"]
    dims = f["dims"]
    p = [theta[i] for i in f["p"]]
    shape = (X.shape[0], X2.shape[0])
    if t == "se":
        var, ell = p
        r2 = _sqdist(X, X2, dims)
        e = np.exp(-0.5 * r2 / (ell * ell))
        return var * e, [e, var * e * r2 / ell**3]
    if t in ("m12", "m32", "m52"):
        var, ell = p
        r = np.sqrt(_sqdist(X, X2, dims))
        s = r / ell
        if t == "m12":
            e = np.exp(-s)
            return var * e, [e, var * e * s / ell]
        if t == "m32":
            e = np.exp(-SQRT3 * s)
            g = (1.0 + SQRT3 * s) * e
            return var * g, [g, var * 3.0 * s * s * e / ell]
        e = np.exp(-SQRT5 * s)
        g = (1.0 + SQRT5 * s + (5.0 / 3.0) * s * s) * e
        return var * g, [g, var * (5.0 / 3.0) * s * s * (1.0 + SQRT5 * s) * e / ell]
    if t == "periodic":
        # gpflow.kernels.Periodic(SquaredExponential): r = pi (x-x')/p,
        # k = var * exp(-0.5 * sum_d (sin(r_d)/ell)^2)  [3P]
        var, ell, per = p
        s2 = np.zeros(shape)
        sc = np.zeros(shape)  # sum_d d_d * sin(r_d) cos(r_d)
        for d in dims:
            diff = X[:, d][:, None] - X2[:, d][None, :]
            r = np.pi * diff / per
            sn = np.sin(r)
            s2 += sn * sn
            sc += diff * sn * np.cos(r)
        e = np.exp(-0.5 * s2 / (ell * ell))
        k = var * e
        return k, [e, k * s2 / ell**3, k * np.pi * sc / (per * per * ell * ell)]
    if t == "lin":
        (var,) = p
        xx = np.zeros(shape)
        for d in dims:
            xx += X[:, d][:, None] * X2[:, d][None, :]
        return var * xx, [xx]
    if t == "poly":
        var, off = p
        deg = int(f["degree"])
        xx = np.zeros(shape)
        for d in dims:
            xx += X[:, d][:, None] * X2[:, d][None, :]
        base = var * xx + off
        dk = deg * base ** (deg - 1)
        return base**deg, [dk * xx, dk]
    if t == "const":
        (var,) = p
        one = np.ones(shape)
        return var * one, [one]
    if t == "cat":
        (var,) = p
        d = dims[0]
        # tf.round = round-half-to-even = np.rint; compared as int64
        a = np.rint(X[:, d]).astype(np.int64)
        b = np.rint(X2[:, d]).astype(np.int64)
        m = (a[:, None] == b[None, :]).astype(np.float64)
        return var * m, [m]
    if t == "empty":
        return np.zeros(shape), []
    raise ValueError(f"unknown factor type {t!r}")


def gram(spec, theta, X, X2=None):
    """K = sum_t prod_f k_f  (gpflow Sum / Product, regularization.py:112,139,166,181)."""
    X = np.asarray(X, dtype=np.float64)
    X2 = X if X2 is None else np.asarray(X2, dtype=np.float64)
    K = np.zeros((X.shape[0], X2.shape[0]))
    for term in spec["terms"]:
        prod = np.ones_like(K)
        for f in term:
            k, _ = factor_value_and_grads(f, theta, X, X2)
            prod = prod * k
        K += prod
    return K


def gram_param_grads(spec, theta, X):
    """dict param_index -> dK/dtheta_p (dense). Parameters shared between factors add up."""
    X = np.asarray(X, dtype=np.float64)
    out = {}
    for term in spec["terms"]:
        vals = [factor_value_and_grads(f, theta, X) for f in term]
        for gi, f in enumerate(term):
            others = np.ones((X.shape[0], X.shape[0]))
            for fi, (k, _) in enumerate(vals):
                if fi != gi:
                    others = others * k
            for pi, dk in zip(f["p"], vals[gi][1]):
                out[pi] = out.get(pi, 0.0) + others * dk
    return out


# --------------------------------------------------------------------------
# exact log marginal likelihood (model_types_DEPR.py:49-56) and gradient
# --------------------------------------------------------------------------
def _resid(spec, theta, y):
    c = theta[spec["mean"]] if spec.get("mean") is not None else 0.0
    return np.asarray(y, dtype=np.float64).reshape(-1) - c


def lml(spec, theta, X, y):
    """log N(y | c, K + sigma^2 I) = -1/2 |L^-1 r|^2 - sum log L_ii - N/2 log 2 pi."""
    X = np.asarray(X, dtype=np.float64)
    N = X.shape[0]
    Ky = gram(spec, theta, X)
    Ky[np.diag_indices(N)] += theta[spec["noise"]]
    if not np.all(np.isfinite(Ky)):
        raise np.linalg.LinAlgError("non-finite Gram matrix")
    L = np.linalg.cholesky(Ky)
    z = scipy.linalg.solve_triangular(L, _resid(spec, theta, y), lower=True)
    return float(-0.5 * z @ z - np.sum(np.log(np.diag(L))) - 0.5 * N * LOG_2PI)


def lml_and_grad(spec, theta, X, y):
    """(LML, dLML/dtheta over ALL params, constrained coordinates).

    dLML/dtheta_p = 1/2 tr(W dK/dtheta_p), W = alpha alpha^T - Ky^-1;
    dLML/dsigma^2 = 1/2 tr(W); dLML/dc = 1^T alpha  (SURVEY.md Appendix A).
    K_y^-1 comes from LAPACK dpotri on the Cholesky factor and the traces are accumulated term by term
    (no dK/dtheta matrix outlives its term), which is how a competent CPU implementation would do it."""
    X = np.asarray(X, dtype=np.float64)
    N = X.shape[0]
    # every factor is evaluated once; K_y is assembled from the same values the gradient uses
    term_vals = [[factor_value_and_grads(f, theta, X) for f in term] for term in spec["terms"]]
    Ky = np.zeros((N, N))
    for vals in term_vals:
        prod = vals[0][0] if len(vals) == 1 else np.prod([k for k, _ in vals], axis=0)
        Ky += prod
    Ky[np.diag_indices(N)] += theta[spec["noise"]]
    if not np.all(np.isfinite(Ky)):  # (a lengthscale driven to 0 by a long trial step: 0/0) - same outcome as "not
        raise np.linalg.LinAlgError("non-finite Gram matrix")  # positive definite": objective() reports +inf
    L = np.linalg.cholesky(Ky)
    r = _resid(spec, theta, y)
    z = scipy.linalg.solve_triangular(L, r, lower=True)
    alpha = scipy.linalg.solve_triangular(L, z, lower=True, trans="T")
    Kinv, info = scipy.linalg.lapack.dpotri(L, lower=1)
    if info != 0:
        raise np.linalg.LinAlgError("dpotri failed")
    Kinv = np.tril(Kinv) + np.tril(Kinv, -1).T
    W = np.outer(alpha, alpha) - Kinv
    val = float(-0.5 * z @ z - np.sum(np.log(np.diag(L))) - 0.5 * N * LOG_2PI)
    g = np.zeros(len(spec["params"]))
    for term, vals in zip(spec["terms"], term_vals):
        for gi, f in enumerate(term):
            Wo = W
            for fi, (k, _) in enumerate(vals):
                if fi != gi:
                    Wo = Wo * k
            for pi, dk in zip(f["p"], vals[gi][1]):
                g[pi] += 0.5 * float(np.vdot(Wo, dk))
    g[spec["noise"]] += 0.5 * np.trace(W)
    if spec.get("mean") is not None:
        g[spec["mean"]] += np.sum(alpha)
    return val, g


# --------------------------------------------------------------------------
# Horseshoe log-density: TFP `Horseshoe._log_prob` [3P, restated from the
# published approximation; PARITY UNPINNED - no stored value in the reference].
# Call site: waveome/model_classes.py:857-859 (scale = 1/penalization_factor).
# --------------------------------------------------------------------------
_HS_G = 0.5614594835668851  # exp(-EulerGamma)
_HS_B = 1.0420764938351215  # sqrt(2 (1-g) / (g (2-g)))
_HS_HINF = 1.0801359952503342  # (1-g)(g^2-6g+12) / (3 g (2-g)^2 b)
_HS_POW = 1.0919284281983377


def horseshoe_logpdf(x, scale):
    x = float(x)
    g, b, h_inf = _HS_G, _HS_B, _HS_HINF
    xx = (x / scale) ** 2 / 2.0
    q = 20.0 / 47.0 * xx**_HS_POW
    h = 1.0 / (1.0 + xx**1.5) + h_inf * q / (1.0 + q)
    c = -0.5 * math.log(2.0 * math.pi**3) - math.log(g * scale)
    z = math.log1p(-g) - math.log(g)
    sp = float(np.logaddexp(0.0, z - xx / (1.0 - g)))
    return -sp + math.log(math.log1p(g / xx - (1.0 - g) / (h + b * xx) ** 2)) + c


def horseshoe_dlogpdf(x, scale):
    """d/dx of horseshoe_logpdf (analytic)."""
    x = float(x)
    g, b, h_inf = _HS_G, _HS_B, _HS_HINF
    xx = (x / scale) ** 2 / 2.0
    dxx = x / (scale * scale)
    q = 20.0 / 47.0 * xx**_HS_POW
    dq = 20.0 / 47.0 * _HS_POW * xx ** (_HS_POW - 1.0)
    h = 1.0 / (1.0 + xx**1.5) + h_inf * q / (1.0 + q)
    dh = -1.5 * math.sqrt(xx) / (1.0 + xx**1.5) ** 2 + h_inf * dq / (1.0 + q) ** 2
    z = math.log1p(-g) - math.log(g)
    t = z - xx / (1.0 - g)
    # d/dxx of -softplus(t) = sigmoid(t) / (1-g)
    d1 = float(sigmoid(t)) / (1.0 - g)
    den = h + b * xx
    inner = g / xx - (1.0 - g) / den**2
    l1p = math.log1p(inner)
    # d2 * dxx with numerator and denominator multiplied by xx^2 (xx^2 underflows for x < 1e-77):
    #   dinner / (1 + inner) = (-g + xx^2 c1) / (xx (xx + g - xx (1-g)/den^2)),  dxx / xx = 2 / x
    c1 = 2.0 * (1.0 - g) * (dh + b) / den**3
    d2dxx = (-g + xx * xx * c1) / ((xx + g - xx * (1.0 - g) / den**2) * l1p) * (2.0 / x)
    return d1 * dxx + d2dxx


def log_prior(spec, theta):
    """sum over trainable params carrying a Horseshoe prior (model_classes.py:862-864)."""
    tot = 0.0
    for i, p in enumerate(spec["params"]):
        if p["trainable"] and p.get("prior_scale") is not None:
            tot += horseshoe_logpdf(theta[i], p["prior_scale"])
    return tot


def log_prior_grad(spec, theta):
    g = np.zeros(len(spec["params"]))
    for i, p in enumerate(spec["params"]):
        if p["trainable"] and p.get("prior_scale") is not None:
            g[i] = horseshoe_dlogpdf(theta[i], p["prior_scale"])
    return g


# --------------------------------------------------------------------------
# training loss in unconstrained coordinates and the SciPy L-BFGS-B fit
# (gpflow `training_loss = -(maximum_log_likelihood_objective + log_prior_density)`,
#  gpflow.optimizers.Scipy -> scipy.optimize.minimize(jac=True, method="L-BFGS-B"))
# --------------------------------------------------------------------------
def objective(spec, u, X, y):
    """(loss, dloss/du) with loss = -(LML + log prior); u = unconstrained trainables."""
    theta = unpack_u(spec, u)
    try:
        val, g = lml_and_grad(spec, theta, X, y)
    except np.linalg.LinAlgError:
        return np.inf, np.zeros_like(u)
    val += log_prior(spec, theta)
    g = g + log_prior_grad(spec, theta)
    idx = trainable_index(spec)
    gu = np.array(
        [g[i] * dconstrain_one(spec["params"][i]["transform"], u[k]) for k, i in enumerate(idx)]
    )
    return -val, -gu


def fit(spec, X, y, maxiter=50000, maxfun=None, ftol=2.220446049250313e-09, gtol=1e-05, maxcor=10,
        maxls=20, u0=None, callback=None):
    """SciPy L-BFGS-B from GPflow defaults; returns dict(theta, u, loss, lml, nit, nfev).

    maxiter/maxfun = 50000 mirrors waveome/model_classes.py:310-313."""
    if u0 is None:
        u0 = pack_u(spec, theta_init(spec))
    res = scipy.optimize.minimize(
        lambda u: objective(spec, u, X, y),
        u0,
        jac=True,
        method="L-BFGS-B",
        callback=callback,
        options={"maxiter": maxiter, "maxfun": maxfun or maxiter, "ftol": ftol, "gtol": gtol,
                 "maxcor": maxcor, "maxls": maxls},
    )
    theta = unpack_u(spec, res.x)
    try:
        end_lml = lml(spec, theta, X, y)
    except np.linalg.LinAlgError:
        # SciPy's line search does not recover from a +inf trial value (its interpolation turns the iterate into
        # NaN); the reference's TensorFlow Cholesky raises there and waveome retries from random parameters
        # (model_classes.py:323-340).  The checker reports such a fit as failed: no finite score.
        end_lml = -np.inf
    return {
        "theta": theta,
        "u": res.x,
        "loss": float(res.fun) if np.isfinite(end_lml) else np.inf,
        "lml": end_lml,
        "nit": int(res.nit),
        "nfev": int(res.nfev),
        "status": int(res.status),
        "message": str(res.message),
    }


# --------------------------------------------------------------------------
# post-fit quantities on the path: predict_y at the training inputs (model_search.py:340) and the
# leave-one-component-out means of calc_feature_importance_components (utilities.py:614-707)
# --------------------------------------------------------------------------
def predict_train(spec, theta, X, y, include_noise=True):
    """Exact-GP posterior mean / variance of f (or y) at the training inputs."""
    X = np.asarray(X, dtype=np.float64)
    N = X.shape[0]
    K = gram(spec, theta, X)
    s2 = theta[spec["noise"]]
    Ky = K + s2 * np.eye(N)
    c = theta[spec["mean"]] if spec.get("mean") is not None else 0.0
    sol = np.linalg.solve(Ky, np.column_stack([_resid(spec, theta, y), K]))
    mean = c + K @ sol[:, 0]
    var = np.diag(K) - np.einsum("ij,ji->i", K, sol[:, 1:]) + (s2 if include_noise else 0.0)
    return mean, var


def predict(spec, theta, X, y, Xnew, include_noise=True):
    """Exact-GP posterior mean / variance at new inputs (gpflow GPR.predict_f / predict_y, full_cov=False)."""
    X = np.asarray(X, dtype=np.float64)
    Xnew = np.asarray(Xnew, dtype=np.float64)
    N = X.shape[0]
    s2 = theta[spec["noise"]]
    L = np.linalg.cholesky(gram(spec, theta, X) + s2 * np.eye(N))
    Ks = gram(spec, theta, Xnew, X)
    c = theta[spec["mean"]] if spec.get("mean") is not None else 0.0
    z = scipy.linalg.solve_triangular(L, _resid(spec, theta, y), lower=True)
    V = scipy.linalg.solve_triangular(L, Ks.T, lower=True)
    kss = np.array([gram(spec, theta, Xnew[i:i + 1])[0, 0] for i in range(len(Xnew))])
    return c + V.T @ z, kss - np.sum(V * V, axis=0) + (s2 if include_noise else 0.0)


def sub_spec(spec, term_ids):
    out = dict(spec)
    out["terms"] = [spec["terms"][t] for t in term_ids]
    return out


def whitened_means(spec, theta, X, y, term_sets, jitter=1e-6):
    """mean_m = chol(K_m + jitter I) q_mu + c with q_mu = chol(K + jitter I)' alpha: gpflow's whitened
    SVGP conditional with Z = X when the kernel is swapped for a sub-sum and q_mu is kept."""
    X = np.asarray(X, dtype=np.float64)
    N = X.shape[0]
    K = gram(spec, theta, X)
    alpha = np.linalg.solve(K + theta[spec["noise"]] * np.eye(N), _resid(spec, theta, y))
    q_mu = np.linalg.cholesky(K + jitter * np.eye(N)).T @ alpha
    c = theta[spec["mean"]] if spec.get("mean") is not None else 0.0
    out = []
    for ts in term_sets:
        Km = gram(sub_spec(spec, ts), theta, X) if len(ts) else np.zeros((N, N))
        out.append(np.linalg.cholesky(Km + jitter * np.eye(N)) @ q_mu + c)
    return np.array(out)


# --------------------------------------------------------------------------
# small spec builders used by tests / bench (GPflow defaults: all ones, c = 0)
# --------------------------------------------------------------------------
def make_spec(D, terms, mean=True, noise_init=1.0, prior_scale=None):
    """terms: list of list of (type, dims[, opts]) tuples.

    opts: {"freeze_variance": bool, "degree": int, "init": [..]}.
    Kernel variance parameters get the Horseshoe prior when prior_scale is given and
    they are trainable (model_classes.py:862-864: name contains 'kernel' and 'variance')."""
    params = []
    out_terms = []
    npar = {"se": 2, "m12": 2, "m32": 2, "m52": 2, "periodic": 3, "lin": 1, "poly": 2, "const": 1,
            "cat": 1, "empty": 0}
    names = {"se": ["variance", "lengthscales"], "m12": ["variance", "lengthscales"],
             "m32": ["variance", "lengthscales"], "m52": ["variance", "lengthscales"],
             "periodic": ["base_kernel.variance", "base_kernel.lengthscales", "period"],
             "lin": ["variance"], "poly": ["variance", "offset"], "const": ["variance"],
             "cat": ["variance"], "empty": []}
    for ti, term in enumerate(terms):
        ft = []
        for fi, fac in enumerate(term):
            typ, dims = fac[0], list(fac[1])
            opts = fac[2] if len(fac) > 2 else {}
            idx = []
            for k in range(npar[typ]):
                nm = f"kernel.kernels[{ti}].kernels[{fi}].{names[typ][k]}"
                is_var = names[typ][k].endswith("variance")
                trainable = not (is_var and opts.get("freeze_variance", False))
                init = opts.get("init", [1.0] * npar[typ])[k]
                params.append({
                    "name": nm, "init": float(init), "transform": "softplus", "trainable": trainable,
                    "prior_scale": (prior_scale if (is_var and trainable) else None),
                })
                idx.append(len(params) - 1)
            ft.append({"type": typ, "dims": dims, "p": idx, "degree": int(opts.get("degree", 3))})
        out_terms.append(ft)
    params.append({"name": "likelihood.variance", "init": float(noise_init),
                   "transform": "softplus_shift", "trainable": True, "prior_scale": None})
    noise = len(params) - 1
    mean_idx = None
    if mean:
        params.append({"name": "mean_function.c", "init": 0.0, "transform": "identity",
                       "trainable": True, "prior_scale": None})
        mean_idx = len(params) - 1
    return {"D": int(D), "params": params, "terms": out_terms, "noise": noise, "mean": mean_idx}
