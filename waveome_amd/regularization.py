"""full_kernel_build - the saturated additive kernel of waveome/regularization.py:14-189.

Same argument names, defaults and term order (unit categorical, categoricals, numeric x base
kernels, optional unit x numeric, categorical x numeric (default on, categorical variance frozen),
optional numeric x numeric), with the `ranks` machinery reduced to rank 1 (it only matters to the
multi-output LMC model, which is outside this path).  Also `make_folds` (:245-276), the fold
builder of `PSVGP.penalization_search`."""
from __future__ import annotations

import copy

import numpy as np

from . import kernels as K


def make_folds(X, unit_col, k_fold=5, random_seed=None):
    """Row indices of k folds, sampled per observation or per unit (waveome/regularization.py:245-276):
    same shuffle (`np.random.seed` + `np.random.shuffle`) and the same divmod split, so a seed gives the
    reference's folds."""
    if random_seed is not None:
        np.random.seed(random_seed)
    if unit_col is None:
        sample_idx = np.arange(0, X.shape[0])
    else:
        sample_idx = np.unique(X[:, unit_col])
        assert len(sample_idx) >= k_fold, (
            "Not enough unique units for number of folds requested, "
            f"{len(sample_idx)} unit(s) < {k_fold} fold(s)")
    np.random.shuffle(sample_idx)
    div, mod = divmod(len(sample_idx), k_fold)
    folds = [sample_idx[(i * div + min(i, mod)):((i + 1) * div + min(i + 1, mod))] for i in range(k_fold)]
    if unit_col is not None:
        folds = [np.where(np.isin(X[:, unit_col], f))[0] for f in folds]
    return folds


def full_kernel_build(cat_vars=[], num_vars=[], unit_idx=None, var_names=None, second_order_numeric=False,
                      categorical_numeric_interactions=True, unit_numeric_interactions=False,
                      return_sum=False, kerns=None, num_outputs=None, ranks=None):
    if kerns is None:
        kerns = [K.SquaredExponential()]
    if ranks is not None or num_outputs is not None:
        raise NotImplementedError("ranks / num_outputs belong to the multi-output path (out of scope)")
    kernel_list, var_list = [], []

    def _copy(k, dim):
        k_copy = copy.deepcopy(k)
        k_copy.active_dims = [dim]
        return k_copy

    if unit_idx is not None:
        cat_vars = [x for x in cat_vars if x != unit_idx]
        kernel_list.append(K.Categorical(active_dims=[unit_idx]))
        if var_names is not None:
            var_list.append("categorical[" + var_names[unit_idx] + "]")
    for c in cat_vars:
        kernel_list.append(K.Categorical(active_dims=[c]))
        if var_names is not None:
            var_list.append("categorical[" + var_names[c] + "]")
    for n in num_vars:
        for k in kerns:
            k_copy = _copy(k, n)
            kernel_list.append(k_copy)
            if var_names is not None:
                var_list.append(k_copy.name + "[" + var_names[n] + "]")
    if unit_numeric_interactions is True and unit_idx is not None:
        for n in num_vars:
            for k in kerns:
                k1 = K.Categorical(active_dims=[unit_idx])
                K.set_trainable(k1.variance, False)
                k2 = _copy(k, n)
                kernel_list.append(K.Product([k1, k2]))
                if var_names is not None:
                    var_list.append(f"{k1.name}[{var_names[unit_idx]}]*{k2.name}[{var_names[n]}]")
    if categorical_numeric_interactions is True:
        for c in cat_vars:
            for n in num_vars:
                for k in kerns:
                    k1 = K.Categorical(active_dims=[c])
                    K.set_trainable(k1.variance, False)
                    k2 = _copy(k, n)
                    kernel_list.append(K.Product([k1, k2]))
                    if var_names is not None:
                        var_list.append(f"{k1.name}[{var_names[c]}]*{k2.name}[{var_names[n]}]")
    if second_order_numeric is True:
        n_count = 0
        for n_first in num_vars:
            for k_first in kerns:
                for n_second in num_vars[n_count:]:
                    for k_second in kerns:
                        k1, k2 = _copy(k_first, n_first), _copy(k_second, n_second)
                        kernel_list.append(K.Product([k1, k2]))
                        if var_names is not None:
                            var_list.append(f"{k1.name}[{var_names[n_first]}]*{k2.name}[{var_names[n_second]}]")
            n_count += 1
    out_kernel = K.Sum(kernel_list) if return_sum is True else kernel_list
    return (out_kernel, var_list) if var_names is not None else out_kernel
