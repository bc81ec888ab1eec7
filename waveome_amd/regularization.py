"""full_kernel_build - the saturated additive kernel of waveome/regularization.py:14-189.

Same argument names, defaults and term order (unit categorical, categoricals, numeric x base
kernels, optional unit x numeric, categorical x numeric (default on, categorical variance frozen),
optional numeric x numeric), with the `ranks` machinery reduced to rank 1 (it only matters to the
multi-output LMC model, which is outside this path).  Also `make_folds` (:245-276), the fold
builder of `PSVGP.penalization_search`.

The term ORDER and the label strings are contractual (downstream code indexes `kernel.kernels` and prints the
labels); the construction here is a plan-then-build: first the list of terms as tuples of (prototype, column,
frozen-variance) factor descriptions, then one pass that instantiates kernels and labels from it."""
from __future__ import annotations

import copy
from itertools import product as cartesian

import numpy as np

from . import kernels as K


def make_folds(X, unit_col, k_fold=5, random_seed=None):
    """Row indices of k folds, sampled per observation or - with `unit_col` - per unit, so that a unit's rows never
    straddle folds (waveome/regularization.py:245-276).  The labels (row numbers / unit ids) are permuted by the
    legacy NumPy RNG after `np.random.seed(random_seed)` and cut into k nearly equal consecutive runs, the first
    `len % k` one longer - the reference's shuffle + divmod split, so a seed reproduces its folds."""
    if random_seed is not None:
        np.random.seed(random_seed)
    by_unit = unit_col is not None
    labels = np.unique(X[:, unit_col]) if by_unit else np.arange(X.shape[0])
    assert not by_unit or len(labels) >= k_fold, (
        "Not enough unique units for number of folds requested, "
        f"{len(labels)} unit(s) < {k_fold} fold(s)")
    runs = np.array_split(np.random.permutation(labels), k_fold)
    if not by_unit:
        return runs
    return [np.flatnonzero(np.isin(X[:, unit_col], run)) for run in runs]


def _categorical(dim, frozen=False):
    k = K.Categorical(active_dims=[dim])
    if frozen:  # inside a product the categorical factor's variance is fixed at 1 (regularization.py:109, 132)
        K.set_trainable(k.variance, False)
    return k


def _on_column(prototype, dim):
    k = copy.deepcopy(prototype)
    k.active_dims = [dim]
    return k


def full_kernel_build(cat_vars=[], num_vars=[], unit_idx=None, var_names=None, second_order_numeric=False,
                      categorical_numeric_interactions=True, unit_numeric_interactions=False,
                      return_sum=False, kerns=None, num_outputs=None, ranks=None):
    if ranks is not None or num_outputs is not None:
        raise NotImplementedError("ranks / num_outputs belong to the multi-output path (out of scope)")
    base = [K.SquaredExponential()] if kerns is None else list(kerns)
    has_unit = unit_idx is not None
    cats = [c for c in cat_vars if not (has_unit and c == unit_idx)]
    nums = list(num_vars)
    CAT = "categorical"  # a factor description is (prototype or CAT, column, frozen variance)

    plan = []
    # main effects: unit, other categoricals, every base kernel on every numeric column (:57-95)
    plan += [((CAT, d, False),) for d in ([unit_idx] if has_unit else []) + cats]
    plan += [((proto, n, False),) for n, proto in cartesian(nums, base)]
    # unit x numeric (:103-121, default off) and categorical x numeric (:124-149, default on)
    if unit_numeric_interactions is True and has_unit:
        plan += [((CAT, unit_idx, True), (proto, n, False)) for n, proto in cartesian(nums, base)]
    if categorical_numeric_interactions is True:
        plan += [((CAT, c, True), (proto, n, False)) for c, n, proto in cartesian(cats, nums, base)]
    # numeric x numeric (:152-178, default off): column pairs (i <= j) x kernel pairs, first factor's kernel outermost
    if second_order_numeric is True:
        plan += [((p1, n1, False), (p2, n2, False))
                 for i, n1 in enumerate(nums) for p1 in base for n2 in nums[i:] for p2 in base]

    def build(desc):
        what, dim, frozen = desc
        return _categorical(dim, frozen) if what is CAT else _on_column(what, dim)

    terms, labels = [], []
    for factors in plan:
        ks = [build(f) for f in factors]
        terms.append(ks[0] if len(ks) == 1 else K.Product(ks))
        if var_names is not None:
            labels.append("*".join(f"{k.name}[{var_names[f[1]]}]" for k, f in zip(ks, factors)))
    out_kernel = K.Sum(terms) if return_sum is True else terms
    return (out_kernel, labels) if var_names is not None else out_kernel
