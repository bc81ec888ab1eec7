"""Hot-path subset of waveome/utilities.py, GPflow-free.

calc_bic (:77-95), calc_rsquare (:131-188), check_if_model_exists (:281-307), print_kernel_names (:366-383), find_variance_components (:1012-1062),
keep_kernel_lengthscale_ / search_through_kernel_list_ (:1136-1184),
freeze_variance_parameters (:977-986).  Same names, argument meaning and return shapes."""
from __future__ import annotations

import numpy as np

from . import kernels as K


_precision = "float64"


FLOAT32_LOSS_REL_ERR = 1e-6   # measured: 1.4e-7 at N = 16384, Matern-3/2 (profiles/r05_bench_c5.json); 6e-8 at N = 2048


def set_precision(precision: str = "float64"):
    """Default floating point precision of contexts created afterwards (waveome/utilities.py:34-49).

    "float32": the N x N work matrices and the operands of their MFMA products are float32 (the 128-row leaves, the
    solves and every reduction stay float64), which nearly doubles the matrix rate (C5: 100 TFLOP/s against 60).  Since
    round 5 it is a PARITY mode for well-conditioned problems: the products accumulate through float64 sums (flushed every
    512 k), and the value of an evaluation is corrected back to float64 through the exact identity
    log|K_y| = -2 log|M| + log|I + E|, E = M K_y M' - I, with K_y recomputed in float64 (first order in E from the gradient
    pass at no cost; `WV_F32_ORDER=2|4` adds higher orders).  Measured against the float64 path: loss / LML 1.4e-7 relative
    and gradient 1e-6 of its largest component at N = 16384 (BASELINE C5), <= 3e-7 for cond(K_y) ~ 5e4 at every size tried;
    what is left scales with (cond(K_y) x 6e-8)^2 - an ill-conditioned K_y (cond > 1e6: noise variance 0.01 under unit-variance
    kernels at N > 4000) still loses digits (1e-5 ... 1e-4; 1e-6 ... 1e-5 with WV_F32_ORDER=4).  The north-star tolerance (LML
    within 1e-6 of the float64 reference) is asserted for float32 in tests/test_gpu_f32.py.  Value-only evaluations (`wv_lml`)
    run the gradient pipeline on a float32 context (the correction needs K_y^-1), so they cost what value + gradient costs."""
    global _precision
    if precision not in ["float32", "float64"]:
        raise ValueError("Precision must be either 'float32' or 'float64'")
    _precision = precision


def get_precision():
    return _precision


def restart_points(n_trainable: int, num_restart: int, loc: float = 0.0, scale: float = 1.0, random_seed=None):
    """Start points of `random_restart_optimize` (waveome/model_classes.py:472-524): restart i draws every
    unconstrained value ~ N(loc, scale) under `np.random.seed(seed)`, seed = i when random_seed is None, else
    random_seed + i + 1 (:498-503, 181-234).  Every model of a run gets the same points, as in the reference.
    (The reference's draws also feed q_mu / q_sqrt, which the exact-GP path does not have, so the numbers differ
    from the reference's for the same seed.)"""
    pts = np.empty((num_restart, n_trainable))
    for i in range(num_restart):
        np.random.seed(i if random_seed is None else random_seed + i + 1)
        pts[i] = np.random.normal(loc=loc, scale=scale, size=n_trainable)
    return pts


def calc_bic(loglik: float, n: int, k: int):
    """2k - 2 loglik (the reference's 'BIC' does not use log n, waveome/utilities.py:94-95)."""
    return 2 * k - 2 * loglik


def check_if_model_exists(model_name, model_list):
    """True when a model with the same additive components (product factors in any order) is in
    `model_list` (waveome/utilities.py:281-307; components compare as sorted character strings there too)."""
    target = set("".join(sorted(x)) for x in model_name.split("+"))
    for other in model_list:
        if target == set("".join(sorted(x)) for x in other.split("+")):
            return True
    return False


def calc_rsquare(m, data=None):
    """R-squared attributed to each additive component, then the unexplained share (waveome/utilities.py:131-188).
    The model's R^2 = 1 - SS_res / SS_tot is split over the components by how close each component's own
    prediction comes to the full prediction: weight_k = 1 - d_k / sum_j d_j, d_k = |full mean - mean_k|^2.
    `m.component_means(data, only=True)` supplies [full, component 0 alone, 1 alone, ...] training means."""
    X, Y = m.data if data is None else data
    y = np.asarray(Y, dtype=np.float64).reshape(-1)
    fitted = np.asarray(m.predict_y(X)[0], dtype=np.float64).reshape(-1)
    r2 = 1.0 - np.sum(np.square(y - fitted)) / np.sum(np.square(y - y.mean()))
    if m.kernel.name != "sum":
        return [np.round(r2, 3), np.round(1 - r2, 3)]
    alone = np.asarray(m.component_means(data, only=True), dtype=np.float64).reshape(-1, len(y))[1:]
    gap = np.sum(np.square(fitted[None, :] - alone[:len(m.kernel.kernels)]), axis=1)
    return [np.round(r2 * (1.0 - g / gap.sum()), 3) for g in gap] + [np.round(1 - r2, 3)]


def replace_kernel_variables(k_name, col_names):
    """Indexed kernel names -> column names: "[i]" becomes "[col_names[i]]" (waveome/utilities.py:262-278).
    Replacements are applied in column order, one after the other, as the reference does."""
    import functools

    return functools.reduce(lambda name, ic: name.replace(f"[{ic[0]}]", f"[{ic[1]}]"), enumerate(col_names), k_name)


def _parts(kernel):
    """Children of a Sum / Product node, None for a base kernel."""
    return getattr(kernel, "kernels", None)


def print_kernel_names(kernel, with_idx=False):
    """Printable structure (waveome/utilities.py:366-383): a base kernel is its name (with "[column]" when
    with_idx), a product joins its factors with "*", a sum is the LIST of its components' names."""
    if kernel is None:
        return ""
    children = _parts(kernel)
    if children is None:
        return f"{kernel.name}[{kernel.active_dims[0]}]" if with_idx else kernel.name
    labels = [print_kernel_names(child, with_idx) for child in children]
    if kernel.name == "product":
        return "*".join(labels)
    return labels if kernel.name == "sum" else []


def freeze_variance_parameters(kernel):
    if kernel.name in ["sum", "product"]:
        for k in kernel.kernels:
            freeze_variance_parameters(k)
    elif kernel.name == "periodic":
        freeze_variance_parameters(kernel.base_kernel)
    elif hasattr(kernel, "variance"):
        K.set_trainable(kernel.variance, False)


def _variance_of(kern, product_scale=1.0):
    """Signal variance a kernel (sub)tree contributes: a base kernel's `variance` (Periodic: its base kernel's,
    Empty: 0), a product multiplies its factors', a nested sum adds its components'."""
    if kern.name == "product":
        return product_scale * float(np.prod([_variance_of(f) for f in kern.kernels]))
    if kern.name == "sum":
        return float(sum(_variance_of(c) for c in kern.kernels))
    if kern.name == "empty":
        return 0.0
    holder = kern.base_kernel if kern.name == "periodic" else kern
    return float(holder.variance.numpy())


def find_variance_components(kern, sum_reduce=True, penalize_factor_prod=1, return_numpy=True):
    """Variance of every additive component (waveome/utilities.py:1012-1062).  A Sum gives one row per component
    (shape [T, 1]) or their total with sum_reduce; anything else a length-1 array.  `penalize_factor_prod` scales
    a top-level product only, as in the reference (its recursion does not pass it down)."""
    if kern.name == "sum":
        rows = np.array([[_variance_of(component)] for component in kern.kernels])
        return rows.sum() if sum_reduce else rows
    return np.array([_variance_of(kern, penalize_factor_prod)])


def keep_kernel_lengthscale_(kernel_component, X):
    """Lengthscale rule of the post-fit pruning (waveome/utilities.py:1136-1153): a factor survives unless its
    lengthscale is at least three times the spread (max - min) of the column it acts on; kernels without a
    lengthscale always survive."""
    holder = kernel_component.base_kernel if kernel_component.name == "periodic" else kernel_component
    ell = getattr(holder, "lengthscales", None)
    if ell is None:
        return True
    column = np.asarray(X)[:, holder.active_dims[0]]
    return bool(ell.numpy() < 3.0 * (column.max() - column.min()))


def search_through_kernel_list_(kernel_list, list_type="sum", X=None):
    """Apply the lengthscale rule to the members of a Sum (or, recursively, a Product) and re-assemble what is
    left (waveome/utilities.py:1156-1184): nothing left -> Empty(), one member -> that member, else a new
    Sum / Product.  A product is replaced by its own pruned version (possibly Empty) and always kept."""
    survivors = [search_through_kernel_list_(k.kernels, "product", X) if k.name == "product" else k
                 for k in kernel_list if k.name == "product" or keep_kernel_lengthscale_(k, X)]
    if not survivors:
        return K.Empty()
    if len(survivors) == 1:
        return survivors[0]
    return (K.Sum if list_type == "sum" else K.Product)(survivors)


def _gaussian_logdensity(x, mu, var):
    """gpflow.logdensities.gaussian."""
    return -0.5 * (np.log(2 * np.pi) + np.log(var) + np.square(mu - x) / var)


def calc_deviance_explained(model, data=None, model_mu=None, model_var=None, base_mu=None, base_var=None,
                            aggregate=True, return_deviance_explained=True, return_loglik=False):
    """Gaussian branch of waveome/utilities.py:517-611 (the likelihood on this path).  All three reference
    densities share the variance var(y), so the saturated model has zero residual and the deviances are scaled
    squared residuals: 2 (sat_ll - ll(mu)) = (y - mu)^2 / var(y) row by row."""
    y = np.asarray(data[1], dtype=np.float64)
    s2 = np.var(y)
    centre = np.mean(y) if base_mu is None else base_mu
    null_dev = np.square(y - centre) / s2
    model_dev = np.square(y - model_mu) / s2
    if return_loglik:
        sat_ll = np.full(np.broadcast(y, model_mu).shape, -0.5 * np.log(2 * np.pi * s2))
        return sat_ll - 0.5 * null_dev, sat_ll - 0.5 * model_dev, sat_ll
    if aggregate:
        null_dev, model_dev = float(np.sum(null_dev)), float(np.sum(model_dev))
        explained = 1 - model_dev / null_dev if null_dev > 0 else 0
    else:
        # rows whose null deviance is 0 (y equals the base mean) explain nothing: ratio 1 there, as in the
        # reference (utilities.py:600-607)
        ratio = np.ones_like(model_dev, dtype=float)
        np.divide(model_dev, null_dev, out=ratio, where=null_dev != 0)
        explained = 1 - ratio
    return explained if return_deviance_explained else (null_dev, model_dev)


def calc_feature_importance_components(model, data=None, return_value="log_bf"):
    """Importance of every additive component, then the residual share (waveome/utilities.py:614-707).

    The reference pops one component at a time from a copy of the SVGP (keeping q_mu), predicts, and compares
    Gaussian log-densities that all use var(y); `model.component_means(data)` returns those predictions in one
    batched GPU call - row 0 the full model, row 1 + k the model without component k.  With a common variance the
    log-densities differ only through residual sums of squares, so with R(.) = sum (y - mean)^2:
        log_bf_k    = (R(without k) - R(full)) / (2 var y)             rounded to 1 decimal
        statistic_k = max(2 log_bf_k, 0)                               rounded to 1 decimal
        de_k        = 1 - (R(without k) - R(full)) / (R(null) - R(full)) clipped to [0, 1], 3 decimals
    and the last entry is 1 - deviance explained = R(full) / R(null) (0 explained if the order
    saturated >= model >= null is violated).  A single-component model is compared with the null model; a lone
    Constant kernel scores 0."""
    y = np.asarray(data[1], dtype=np.float64).reshape(-1)
    means = np.asarray(model.component_means(data), dtype=np.float64).reshape(-1, len(y))
    half_inv_var = 0.5 / np.var(y)
    rss = np.sum(np.square(means - y[None, :]), axis=1)   # [full, without 0, without 1, ...]
    rss_null = float(np.sum(np.square(y - y.mean())))
    rss_full = float(rss[0])
    explained = min(1.0, max(0.0, 1.0 - rss_full / rss_null)) if rss_full <= rss_null else 0.0

    def score(rss_without):
        gain = (rss_without - rss_full) * half_inv_var       # log Bayes factor of the full model over the reduced
        if return_value == "log_bf":
            return np.round(gain, 1)
        if return_value == "statistic":
            return max(np.round(2.0 * gain, 1), 0)
        share = 1.0 - (rss_without - rss_full) / (rss_null - rss_full)
        return np.round(min(1.0, max(0.0, share)), 3)

    kernel = model.kernel
    if kernel.name == "sum":
        scores = [score(float(r)) for r in rss[1:1 + len(kernel.kernels)]]
    elif kernel.name == "constant":
        scores = [0.0]
    elif return_value in ("log_bf", "statistic"):
        gain = (rss_null - rss_full) * half_inv_var
        scores = [np.round(gain if return_value == "log_bf" else 2.0 * gain, 1)]
    else:
        scores = [np.round(explained, 3)]
    return scores + [np.round(1 - explained, 3)]
