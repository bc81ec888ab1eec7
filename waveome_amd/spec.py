"""Flatten a (kernel, likelihood, mean) model description into the C-ABI's wv_kernel_spec.

K = sum_t prod_f k_f: `Sum` nodes become terms, `Product` nodes become the factors of a
term (products of sums are distributed, which makes the shared parameters appear in several
terms - the device gradient accumulates per parameter index).  Parameter table order is
kernel parameters in tree order, then the likelihood variance, then the mean constant.
Reference structure being encoded: waveome/regularization.py:14-189 (`full_kernel_build`).
"""
from __future__ import annotations

import ctypes as C
import itertools
from typing import List, Optional

import numpy as np

from . import _lib
from . import kernels as K

_TYPE_CODE = {"se": _lib.K_SE, "m12": _lib.K_M12, "m32": _lib.K_M32, "m52": _lib.K_M52,
              "periodic": _lib.K_PERIODIC, "lin": _lib.K_LIN, "poly": _lib.K_POLY,
              "const": _lib.K_CONST, "cat": _lib.K_CAT, "empty": _lib.K_EMPTY}
_TRANSFORM_CODE = {"identity": _lib.T_IDENTITY, "softplus": _lib.T_SOFTPLUS,
                   "softplus_shift": _lib.T_SOFTPLUS_SHIFT}


def _expand(kernel) -> List[List[K.Kernel]]:
    """kernel tree -> list of terms, each a list of base kernels."""
    if isinstance(kernel, K.Sum):
        out = []
        for k in kernel.kernels:
            out.extend(_expand(k))
        return out
    if isinstance(kernel, K.Product):
        parts = [_expand(k) for k in kernel.kernels]
        return [list(itertools.chain.from_iterable(combo)) for combo in itertools.product(*parts)]
    return [[kernel]]


def _named_parameters(kernel, prefix="kernel"):
    """(path, Parameter) pairs with gpflow-style dotted paths."""
    if isinstance(kernel, (K.Sum, K.Product)):
        for i, k in enumerate(kernel.kernels):
            yield from _named_parameters(k, f"{prefix}.kernels[{i}]")
    elif isinstance(kernel, K.Periodic):
        yield f"{prefix}.base_kernel.variance", kernel.base_kernel.variance
        yield f"{prefix}.base_kernel.lengthscales", kernel.base_kernel.lengthscales
        yield f"{prefix}.period", kernel.period
    else:
        for nm in ("variance", "lengthscales", "offset"):
            p = getattr(kernel, nm, None)
            if isinstance(p, K.Parameter):
                yield f"{prefix}.{nm}", p


def _factor_params(k):
    t = k._ptype
    if t in ("se", "m12", "m32", "m52"):
        return [k.variance, k.lengthscales]
    if t == "periodic":
        return [k.base_kernel.variance, k.base_kernel.lengthscales, k.period]
    if t == "poly":
        return [k.variance, k.offset]
    if t in ("lin", "const", "cat"):
        return [k.variance]
    if t == "empty":
        return []
    raise TypeError(f"kernel {k!r} is not supported on the GPU path")


class ModelSpec:
    """Host mirror of wv_kernel_spec, plus the Parameter objects it was built from."""

    def __init__(self, kernel, n_dims: int, likelihood: Optional[K.GaussianLikelihood] = None,
                 mean_function=None):
        self.kernel = kernel
        self.likelihood = likelihood if likelihood is not None else K.GaussianLikelihood()
        self.mean_function = mean_function
        self.n_dims = int(n_dims)
        named = list(_named_parameters(kernel))
        self.param_objs: List[K.Parameter] = []
        self.param_names: List[str] = []
        ids = {}
        for path, p in named:
            if id(p) not in ids:
                ids[id(p)] = len(self.param_objs)
                self.param_objs.append(p)
                self.param_names.append(path)
        self.terms = []
        for term in _expand(kernel):
            ft = []
            for k in term:
                dims = k.active_dims
                if dims is None:
                    dims = [0] if k._ptype in ("cat", "poly") else list(range(self.n_dims))
                if k._ptype in ("cat", "poly") or isinstance(k, K.Lin):
                    dims = [dims[0]]  # single-column kernels (kernels.py:20-23, 100-103)
                ft.append({"type": k._ptype, "dims": [int(d) for d in dims],
                           "p": [ids[id(p)] for p in _factor_params(k)],
                           "degree": int(getattr(k, "degree", 3))})
            self.terms.append(ft)
        self.noise = len(self.param_objs)
        self.param_objs.append(self.likelihood.variance)
        self.param_names.append("likelihood.variance")
        self.mean = None
        if isinstance(mean_function, K.ConstantMean):
            self.mean = len(self.param_objs)
            self.param_objs.append(mean_function.c)
            self.param_names.append("mean_function.c")
        elif mean_function is not None and not isinstance(mean_function, K.ZeroMean):
            raise TypeError("mean_function must be ConstantMean, ZeroMean or None")
        for f in itertools.chain.from_iterable(self.terms):
            for d in f["dims"]:
                if not 0 <= d < self.n_dims:
                    raise ValueError(f"active dim {d} outside the {self.n_dims} columns of X")
        self._keep = None

    # ---- parameter vectors -------------------------------------------------
    @property
    def n_params(self):
        return len(self.param_objs)

    @property
    def trainable_index(self):
        return [i for i, p in enumerate(self.param_objs) if p.trainable]

    @property
    def n_trainable(self):
        return len(self.trainable_index)

    def theta(self) -> np.ndarray:
        return np.array([p.value for p in self.param_objs], dtype=np.float64)

    def assign_theta(self, theta):
        for p, v in zip(self.param_objs, np.asarray(theta, dtype=np.float64)):
            p.assign(v)

    def prior_scale(self, p) -> float:
        return float(p.prior.scale) if isinstance(p.prior, K.Horseshoe) else -1.0

    # ---- exports -------------------------------------------------------------
    def to_dict(self) -> dict:
        """Plain-data form (the format oracle/gp_oracle.py documents)."""
        params = []
        for name, p in zip(self.param_names, self.param_objs):
            ps = self.prior_scale(p)
            params.append({"name": name, "init": p.value, "transform": p.transform,
                           "trainable": p.trainable, "prior_scale": ps if ps > 0 else None})
        return {"D": self.n_dims, "params": params, "terms": [[dict(f) for f in t] for t in self.terms],
                "noise": self.noise, "mean": self.mean}

    # what one kernel structure may hold on the device (include/waveome_amd.h)
    LIMITS = {"columns": 32, "additive terms": 64, "factors": 128, "factors in one product": 4, "parameters": 256}

    def check_limits(self):
        """Raise NotImplementedError, naming the limit, for a structure the device path cannot hold - before the C-ABI
        answers with its terser code (the reference would build such a kernel; nothing is ever truncated here)."""
        have = {"columns": self.n_dims, "additive terms": len(self.terms),
                "factors": sum(len(t) for t in self.terms),
                "factors in one product": max((len(t) for t in self.terms), default=0), "parameters": self.n_params}
        over = [f"{have[k]} {k} (limit {v})" for k, v in self.LIMITS.items() if have[k] > v]
        if over:
            raise NotImplementedError("kernel structure beyond what one device structure holds: " + ", ".join(over)
                                      + " - fit fewer interaction terms at a time (e.g. second_order_numeric=False, "
                                        "fewer base kernels)")

    def to_ctypes(self) -> _lib.KernelSpec:
        self.check_limits()
        n = self.n_params
        params = (_lib.Param * n)()
        for i, p in enumerate(self.param_objs):
            params[i].init = p.value
            params[i].prior_scale = self.prior_scale(p)
            params[i].transform = _TRANSFORM_CODE[p.transform]
            params[i].trainable = 1 if p.trainable else 0
        flat = list(itertools.chain.from_iterable(self.terms))
        factors = (_lib.Factor * max(1, len(flat)))()
        for i, f in enumerate(flat):
            factors[i].type = _TYPE_CODE[f["type"]]
            mask = 0
            for d in f["dims"]:
                mask |= 1 << d
            factors[i].dim_mask = mask
            for k in range(3):
                factors[i].param[k] = f["p"][k] if k < len(f["p"]) else -1
            factors[i].degree = f["degree"]
        starts = (C.c_int32 * (len(self.terms) + 1))()
        acc = 0
        for t, term in enumerate(self.terms):
            starts[t] = acc
            acc += len(term)
        starts[len(self.terms)] = acc
        spec = _lib.KernelSpec()
        spec.n_dims = self.n_dims
        spec.n_params = n
        spec.params = params
        spec.n_terms = len(self.terms)
        spec.term_start = starts
        spec.factors = factors
        spec.noise_param = self.noise
        spec.mean_param = -1 if self.mean is None else self.mean
        self._keep = (params, factors, starts)  # keep the arrays alive with the spec
        return spec

    # ---- host helpers routed through the C-ABI (same code the device path uses) ----
    def block_column(self) -> int:
        """The column whose Categorical factor every non-zero additive term carries, or -1: for such
        a structure K_y is block diagonal in the labels of that column and the device products skip the zero blocks."""
        col = C.c_int32(-2)
        _lib.check(_lib.load().wv_spec_block_column(C.byref(self.to_ctypes()), C.byref(col)), "wv_spec_block_column")
        return int(col.value)

    def constrain(self, u) -> np.ndarray:
        lib = _lib.load()
        u = np.ascontiguousarray(u, dtype=np.float64)
        th = np.empty(self.n_params)
        _lib.check(lib.wv_spec_constrain(C.byref(self.to_ctypes()), _lib.as_dp(u), _lib.as_dp(th)),
                   "wv_spec_constrain")
        return th

    def unconstrain(self, theta=None) -> np.ndarray:
        lib = _lib.load()
        theta = np.ascontiguousarray(self.theta() if theta is None else theta, dtype=np.float64)
        u = np.empty(self.n_trainable)
        _lib.check(lib.wv_spec_unconstrain(C.byref(self.to_ctypes()), _lib.as_dp(theta), _lib.as_dp(u)),
                   "wv_spec_unconstrain")
        return u

    def log_prior(self, theta=None, grad=False):
        lib = _lib.load()
        theta = np.ascontiguousarray(self.theta() if theta is None else theta, dtype=np.float64)
        val = C.c_double()
        g = np.zeros(self.n_params)
        _lib.check(lib.wv_spec_log_prior(C.byref(self.to_ctypes()), _lib.as_dp(theta), C.byref(val),
                                         _lib.as_dp(g)), "wv_spec_log_prior")
        return (val.value, g) if grad else val.value
