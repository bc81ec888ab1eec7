"""Hot-path subset of waveome/utilities.py, GPflow-free.

calc_bic (:77-95), calc_rsquare (:131-188), check_if_model_exists (:281-307), print_kernel_names (:366-383), find_variance_components (:1012-1062),
keep_kernel_lengthscale_ / search_through_kernel_list_ (:1136-1184),
freeze_variance_parameters (:977-986).  Same names, argument meaning and return shapes."""
from __future__ import annotations

import numpy as np

from . import kernels as K


_precision = "float64"


def set_precision(precision: str = "float64"):
    """Default floating point precision of contexts created afterwards (waveome/utilities.py:34-49):
    "float32" keeps the N x N work matrices and their MFMA products in float32."""
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
    """R-squared share of each additive component, then the residual share (waveome/utilities.py:131-188).
    `m.component_means(data, only=True)` supplies [full, component 0 alone, 1 alone, ...] training means."""
    X, Y = m.data if data is None else data
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, 1)
    sse = np.sum((Y - Y.mean()) ** 2)
    mu_all_hat, _ = m.predict_y(X)
    total_rsq = 1 - (np.sum((Y - mu_all_hat) ** 2) / sse)
    rsq = []
    k = m.kernel
    if k.name == "sum":
        means = m.component_means(data, only=True)
        ssr_list = [np.sum((mu_all_hat - means[1 + i].reshape(-1, 1)) ** 2) for i in range(len(k.kernels))]
        for i in range(len(k.kernels)):
            rsq += [np.round(total_rsq * (1 - ssr_list[i] / sum(ssr_list)), 3)]
    else:
        rsq += [np.round(total_rsq, 3)]
    rsq += [np.round(1 - total_rsq, 3)]
    return rsq


def replace_kernel_variables(k_name, col_names):
    """Indexed kernel names -> column names: "[i]" becomes "[col_names[i]]" (waveome/utilities.py:262-278)."""
    new_k_name = k_name
    for i, c in enumerate(col_names):
        new_k_name = new_k_name.replace("[" + str(i) + "]", "[" + c + "]")
    return new_k_name


def print_kernel_names(kernel, with_idx=False):
    names = []
    if kernel is None:
        return ""
    if hasattr(kernel, "kernels") is False:
        if with_idx:
            return kernel.name + "[" + str(kernel.active_dims[0]) + "]"
        return kernel.name
    elif kernel.name == "sum":
        return [print_kernel_names(x, with_idx) for x in kernel.kernels]
    elif kernel.name == "product":
        return "*".join([print_kernel_names(x, with_idx) for x in kernel.kernels])
    return names


def freeze_variance_parameters(kernel):
    if kernel.name in ["sum", "product"]:
        for k in kernel.kernels:
            freeze_variance_parameters(k)
    elif kernel.name == "periodic":
        freeze_variance_parameters(kernel.base_kernel)
    elif hasattr(kernel, "variance"):
        K.set_trainable(kernel.variance, False)


def find_variance_components(kern, sum_reduce=True, penalize_factor_prod=1, return_numpy=True):
    """Variance parameter of every additive component (product = product of its factors')."""
    if kern.name == "sum":
        var_list = np.stack([find_variance_components(kern=x, sum_reduce=sum_reduce) for x in kern.kernels])
        return np.sum(var_list) if sum_reduce else var_list
    elif kern.name == "product":
        return np.array([penalize_factor_prod * np.prod([find_variance_components(x, sum_reduce)
                                                          for x in kern.kernels])])
    elif kern.name == "periodic":
        return np.array([kern.base_kernel.variance.numpy()])
    elif kern.name == "empty":
        return np.zeros(1)
    return np.array([kern.variance.numpy()])


def keep_kernel_lengthscale_(kernel_component, X):
    """False when the lengthscale is >= 3 x the range of its input column (utilities.py:1136-1153)."""
    if kernel_component.name == "periodic":
        kernel_component = kernel_component.base_kernel
    if hasattr(kernel_component, "lengthscales") is False:
        return True
    active_index = kernel_component.active_dims[0]
    var_range = 3 * np.ptp(X[:, active_index])
    return kernel_component.lengthscales.numpy() < var_range


def search_through_kernel_list_(kernel_list, list_type="sum", X=None):
    out_list = []
    for k in kernel_list:
        if k.name == "product":
            out_list.append(search_through_kernel_list_(k.kernels, list_type="product", X=X))
        elif keep_kernel_lengthscale_(k, X):
            out_list.append(k)
    if len(out_list) > 1:
        return K.Sum(out_list) if list_type == "sum" else K.Product(out_list)
    if len(out_list) == 1:
        return out_list[0]
    return K.Empty()


def _gaussian_logdensity(x, mu, var):
    """gpflow.logdensities.gaussian."""
    return -0.5 * (np.log(2 * np.pi) + np.log(var) + np.square(mu - x) / var)


def calc_deviance_explained(model, data=None, model_mu=None, model_var=None, base_mu=None, base_var=None,
                            aggregate=True, return_deviance_explained=True, return_loglik=False):
    """Gaussian branch of waveome/utilities.py:517-611 (the likelihood on this path)."""
    y = np.asarray(data[1], dtype=np.float64)
    y_var = np.var(y)
    sat_ll = _gaussian_logdensity(y, y, y_var)
    base_ll = _gaussian_logdensity(y, np.mean(y) if base_mu is None else base_mu, y_var)
    mod_ll = _gaussian_logdensity(y, model_mu, y_var)
    if return_loglik:
        return base_ll, mod_ll, sat_ll
    if aggregate:
        null_deviance = max(0, 2 * np.sum(sat_ll - base_ll))
        model_deviance = max(0, 2 * np.sum(sat_ll - mod_ll))
        deviance_explained = 1 - (model_deviance / null_deviance) if null_deviance > 0 else 0
    else:
        null_deviance = np.clip(2 * (sat_ll - base_ll), a_min=0, a_max=np.inf)
        model_deviance = np.clip(2 * (sat_ll - mod_ll), a_min=0, a_max=np.inf)
        deviance_explained = 1 - np.divide(model_deviance, null_deviance,
                                           out=np.ones_like(model_deviance, dtype="float"),
                                           where=(null_deviance != 0))
    if return_deviance_explained:
        return deviance_explained
    return null_deviance, model_deviance


def calc_feature_importance_components(model, data=None, return_value="log_bf"):
    """waveome/utilities.py:614-707: per additive component, the fit lost when the component is removed
    from the kernel while the (whitened) variational mean is kept; last entry = 1 - deviance explained.
    `model.component_means(data)` supplies [full, without component 0, 1, ...] training means (GPU)."""
    de_list = []
    k = model.kernel
    means = model.component_means(data)
    y = np.asarray(data[1], dtype=np.float64).reshape(-1, 1)
    full_mu_hat = means[0].reshape(-1, 1)
    null_lls, mod_lls, sat_lls = calc_deviance_explained(model=model, data=(data[0], y), model_mu=full_mu_hat,
                                                         return_deviance_explained=False, aggregate=False,
                                                         return_loglik=True)
    if np.sum(sat_lls) >= np.sum(mod_lls) and np.sum(mod_lls) >= np.sum(null_lls):
        full_de = 1 - (-2 * np.sum(mod_lls - sat_lls) / (-2 * np.sum(null_lls - sat_lls)))
        full_de = max(min(1, full_de), 0)
    else:
        full_de = 0
    if k.name == "sum":
        for k_idx in range(len(k.kernels)):
            mod_mu_hat = means[1 + k_idx].reshape(-1, 1)
            null_lls, sub_mod_lls, sat_lls = calc_deviance_explained(
                model=model, data=(data[0], y), model_mu=mod_mu_hat, return_deviance_explained=False,
                aggregate=False, return_loglik=True)
            if return_value == "statistic":
                scaled_de = max(np.round(-2 * (np.sum(sub_mod_lls) - np.sum(mod_lls)), 1), 0)
            elif return_value == "log_bf":
                scaled_de = np.round(np.sum(mod_lls) - np.sum(sub_mod_lls), 1)
            else:
                scaled_de = 1 - (-2 * np.sum(sub_mod_lls - mod_lls) / (-2 * np.sum(null_lls - mod_lls)))
                scaled_de = np.round(max(min(1, scaled_de), 0), 3)
            de_list += [scaled_de]
    else:
        if k.name == "constant":
            de_list += [0.0]
        elif return_value == "statistic":
            de_list += [np.round(-2 * (np.sum(null_lls) - np.sum(mod_lls)), 1)]
        elif return_value == "log_bf":
            de_list += [np.round((np.sum(mod_lls) - np.sum(null_lls)), 1)]
        else:
            de_list += [np.round(full_de, 3)]
    de_list += [np.round(1 - full_de, 3)]
    return de_list
