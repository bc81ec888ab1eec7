"""Random kernel trees: the flattened Sum-of-Products spec (waveome_amd/spec.py -> wv_kernel_spec, the form the
device interprets) evaluated by the oracle must equal a direct recursive evaluation of the tree with the textbook
kernel formulas - nested sums inside products are distributed, parameters shared by the copies stay shared."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import gp_oracle as o
from waveome_amd import ModelSpec
from waveome_amd import kernels as K

D = 3


def _direct(kern, X):
    """Gram matrix of a kernel tree straight from the definitions (gpflow / waveome formulas)."""
    if kern.name == "sum":
        return sum(_direct(k, X) for k in kern.kernels)
    if kern.name == "product":
        out = np.ones((len(X), len(X)))
        for k in kern.kernels:
            out = out * _direct(k, X)
        return out
    dims = kern.active_dims
    x = X[:, dims]
    if kern.name == "periodic":
        b = kern.base_kernel
        diff = x[:, None, :] - x[None, :, :]
        s = np.sin(np.pi * diff / kern.period.value) / b.lengthscales.value
        return b.variance.value * np.exp(-0.5 * np.sum(s * s, axis=-1))
    if kern.name in ("squared_exponential", "matern12", "matern32", "matern52"):
        r2 = np.sum((x[:, None, :] - x[None, :, :]) ** 2, axis=-1) / kern.lengthscales.value ** 2
        r = np.sqrt(r2)
        v = kern.variance.value
        return {"squared_exponential": v * np.exp(-0.5 * r2), "matern12": v * np.exp(-r),
                "matern32": v * (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r),
                "matern52": v * (1 + np.sqrt(5) * r + 5.0 / 3.0 * r2) * np.exp(-np.sqrt(5) * r)}[kern.name]
    if kern.name == "lin":
        return kern.variance.value * np.outer(x[:, 0], x[:, 0])
    if kern.name == "poly":
        return (kern.variance.value * np.outer(x[:, 0], x[:, 0]) + kern.offset.value) ** kern.degree
    if kern.name == "categorical":
        r = np.rint(x[:, 0])
        return kern.variance.value * (r[:, None] == r[None, :])
    if kern.name == "constant":
        return kern.variance.value * np.ones((len(X), len(X)))
    raise AssertionError(kern.name)


pos = st.floats(min_value=0.3, max_value=2.5)
dim = st.integers(min_value=0, max_value=D - 1)


@st.composite
def base_kernel(draw):
    kind = draw(st.sampled_from(["se", "m12", "m32", "m52", "per", "lin", "poly", "cat", "const"]))
    d = draw(dim)
    if kind in ("se", "m12", "m32", "m52"):
        cls = {"se": K.SquaredExponential, "m12": K.Matern12, "m32": K.Matern32, "m52": K.Matern52}[kind]
        dims = sorted(set([d, draw(dim)]))
        return cls(variance=draw(pos), lengthscales=draw(pos), active_dims=dims)
    if kind == "per":
        return K.Periodic(K.SquaredExponential(variance=draw(pos), lengthscales=draw(pos), active_dims=[d]), period=draw(pos))
    if kind == "lin":
        return K.Lin(active_dims=[d], variance=draw(pos))
    if kind == "poly":
        return K.Poly(active_dims=[d], variance=draw(pos), offset=draw(pos), degree=draw(st.integers(2, 3)))
    if kind == "cat":
        return K.Categorical(active_dims=[2], variance=draw(pos))
    return K.Constant(variance=draw(pos))


def tree(depth):
    if depth == 0:
        return base_kernel()
    sub = st.lists(tree(depth - 1), min_size=2, max_size=3)
    return st.one_of(base_kernel(), sub.map(K.Sum), sub.map(K.Product))


@settings(max_examples=60, deadline=None)
@given(tree(2), st.integers(min_value=0, max_value=10 ** 6))
def test_flattened_spec_equals_direct_tree_evaluation(kern, seed):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(9, D))
    X[:, 2] = rng.integers(0, 3, 9)
    spec = ModelSpec(kern, D, K.GaussianLikelihood(), K.ConstantMean())
    if sum(len(t) for t in spec.terms) > 128 or len(spec.terms) > 64:
        return  # beyond the device limits (wv_spec_check reports that; nothing to compare)
    got = o.gram(spec.to_dict(), spec.theta(), X)
    want = _direct(kern, X)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
