"""Host-side kernel objects for the GP fitting path.

GPflow-free stand-ins with the attribute surface waveome's callers use
(`.name`, `.active_dims`, `.variance`, `.lengthscales`, `.period`, `.base_kernel`,
`.kernels`, `+`, `*`): the custom kernels mirror waveome/kernels.py (`Lin` :5-39,
`Poly` :42-83, `Categorical` :86-124, `Empty` :127-142) and the stationary /
periodic / constant ones mirror the gpflow.kernels classes the reference composes
them with (waveome/regularization.py:23,112,139,166,181).  They hold parameters and
structure only - Gram matrices are evaluated on the GPU from the flattened spec
(waveome_amd/spec.py -> wv_kernel_spec), never on the host.
"""
from __future__ import annotations

import copy
from typing import List, Optional, Sequence

import numpy as np


class Horseshoe:
    """Marker for tfd.Horseshoe(scale) priors (waveome/model_classes.py:857-859)."""

    def __init__(self, scale: float):
        self.scale = float(scale)

    def __repr__(self):
        return f"Horseshoe(scale={self.scale})"


class Parameter:
    """Minimal gpflow.Parameter: constrained value + transform + trainable flag + prior."""

    def __init__(self, value, transform: str = "softplus", trainable: bool = True, prior=None,
                 name: str = ""):
        self.value = float(np.asarray(value).reshape(-1)[0])
        self.transform = transform  # "softplus" | "softplus_shift" | "identity"
        self.trainable = bool(trainable)
        self.prior = prior
        self.name = name

    def numpy(self):
        return np.float64(self.value)

    def assign(self, value):
        self.value = float(np.asarray(value).reshape(-1)[0])

    def __float__(self):
        return self.value

    def __repr__(self):
        return (f"Parameter({self.value!r}, transform={self.transform}, trainable={self.trainable}"
                f", prior={self.prior})")


def positive(value=1.0, **kw) -> Parameter:
    return Parameter(value, transform="softplus", **kw)


def set_trainable(obj, flag: bool):
    """gpflow.utilities.set_trainable for Parameters and kernels."""
    if isinstance(obj, Parameter):
        obj.trainable = bool(flag)
    else:
        for p in obj.parameters:
            p.trainable = bool(flag)


class Kernel:
    name = "kernel"
    _ptype: Optional[str] = None  # factor type tag used by the spec encoder

    def __init__(self, active_dims: Optional[Sequence[int]] = None):
        self.active_dims = None if active_dims is None else [int(d) for d in active_dims]

    @property
    def parameters(self) -> List[Parameter]:
        return [v for v in self.__dict__.values() if isinstance(v, Parameter)]

    @property
    def trainable_parameters(self) -> List[Parameter]:
        return [p for p in self.parameters if p.trainable]

    def __add__(self, other):
        return Sum([self, other])

    def __mul__(self, other):
        return Product([self, other])

    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, copy.deepcopy(v, memo))
        return new

    def __repr__(self):
        return f"{self.name}{self.active_dims if self.active_dims is not None else ''}"


class _Stationary(Kernel):
    def __init__(self, variance=1.0, lengthscales=1.0, active_dims=None):
        super().__init__(active_dims)
        self.variance = positive(variance)
        self.lengthscales = positive(lengthscales)


class SquaredExponential(_Stationary):
    name = "squared_exponential"
    _ptype = "se"


class Matern12(_Stationary):
    name = "matern12"
    _ptype = "m12"


class Matern32(_Stationary):
    name = "matern32"
    _ptype = "m32"


class Matern52(_Stationary):
    name = "matern52"
    _ptype = "m52"


RBF = SquaredExponential


class Periodic(Kernel):
    """gpflow.kernels.Periodic(base_kernel, period); only a SquaredExponential base is
    supported on the GPU path (the only base the reference's defaults use)."""

    name = "periodic"
    _ptype = "periodic"

    def __init__(self, base_kernel: Optional[Kernel] = None, period=1.0):
        base_kernel = SquaredExponential() if base_kernel is None else base_kernel
        if not isinstance(base_kernel, SquaredExponential):
            raise NotImplementedError("Periodic: only a SquaredExponential base kernel is supported")
        super().__init__(base_kernel.active_dims)
        self.base_kernel = base_kernel
        self.period = positive(period)

    @property
    def active_dims(self):
        return self.base_kernel.active_dims

    @active_dims.setter
    def active_dims(self, v):
        if "base_kernel" in self.__dict__:
            self.base_kernel.active_dims = None if v is None else [int(d) for d in v]

    @property
    def parameters(self):
        return self.base_kernel.parameters + [self.period]


class Constant(Kernel):
    name = "constant"
    _ptype = "const"

    def __init__(self, variance=1.0, active_dims=None):
        super().__init__(active_dims)
        self.variance = positive(variance)


class Linear(Kernel):
    """gpflow.kernels.Linear: variance * <x, x'> over active_dims."""

    name = "linear"
    _ptype = "lin"

    def __init__(self, variance=1.0, active_dims=None):
        super().__init__(active_dims)
        self.variance = positive(variance)


class Lin(Kernel):
    """waveome/kernels.py:5-39 - variance * x x' on ONE column (default column 0)."""

    name = "lin"
    _ptype = "lin"

    def __init__(self, active_dims=None, variance=1.0):
        if active_dims is None:
            active_dims = [0]
        super().__init__(active_dims)
        self.variance = positive(variance)
        self.active_index = self.active_dims[0]


class Poly(Kernel):
    """waveome/kernels.py:42-83 - (variance * x x' + offset) ** degree on one column."""

    name = "poly"
    _ptype = "poly"

    def __init__(self, active_dims=(0,), variance=1.0, offset=1.0, degree=3):
        super().__init__(active_dims)
        self.variance = positive(variance)
        self.offset = positive(offset)
        self.degree = int(degree)
        self.active_index = self.active_dims[0]


class Categorical(Kernel):
    """waveome/kernels.py:86-124 - variance * 1[round(x_i) == round(x_j)] on one column."""

    name = "categorical"
    _ptype = "cat"

    def __init__(self, active_dims=(0,), variance=1.0):
        super().__init__(active_dims)
        self.variance = positive(variance)
        self.active_index = self.active_dims[0]


class Empty(Kernel):
    """waveome/kernels.py:127-142 - identically zero; frozen variance 1e-6."""

    name = "empty"
    _ptype = "empty"

    def __init__(self):
        super().__init__([0])
        self.variance = Parameter(1e-6, transform="softplus", trainable=False)


class _Combination(Kernel):
    def __init__(self, kernels: Sequence[Kernel]):
        super().__init__(None)
        flat = []
        for k in kernels:
            if not isinstance(k, Kernel):
                raise TypeError(f"not a kernel: {k!r}")
            # gpflow flattens same-type combinations
            if type(k) is type(self):
                flat.extend(k.kernels)
            else:
                flat.append(k)
        self.kernels = list(flat)

    @property
    def parameters(self):
        out = []
        for k in self.kernels:
            out.extend(k.parameters)
        return out

    def __repr__(self):
        sep = "+" if self.name == "sum" else "*"
        return "(" + sep.join(repr(k) for k in self.kernels) + ")"


class Sum(_Combination):
    name = "sum"


class Product(_Combination):
    name = "product"


# ---- mean functions (gpflow.functions.Constant / Zero) ----
class ConstantMean:
    name = "constant"

    def __init__(self, c=0.0):
        self.c = Parameter(c, transform="identity")

    @property
    def parameters(self):
        return [self.c]


class ZeroMean:
    name = "zero"

    @property
    def parameters(self):
        return []


class GaussianLikelihood:
    """gpflow.likelihoods.Gaussian: variance 1.0, Softplus + Shift(1e-6)."""

    name = "gaussian"

    def __init__(self, variance=1.0):
        self.variance = Parameter(variance, transform="softplus_shift")

    @property
    def parameters(self):
        return [self.variance]
