"""Hot-path subset of waveome/utilities.py, GPflow-free.

calc_bic (:77-95), print_kernel_names (:366-383), find_variance_components (:1012-1062),
keep_kernel_lengthscale_ / search_through_kernel_list_ (:1136-1184),
freeze_variance_parameters (:977-986).  Same names, argument meaning and return shapes."""
from __future__ import annotations

import numpy as np

from . import kernels as K


def calc_bic(loglik: float, n: int, k: int):
    """2k - 2 loglik (the reference's 'BIC' does not use log n, waveome/utilities.py:94-95)."""
    return 2 * k - 2 * loglik


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
