"""The ctypes stub INTEGRATION.md shows a waveome maintainer is real code: this test extracts it from the document and
runs it, verbatim except for the library path, against libwaveome_amd.so - with two shim modules standing in for the
packages this image does not have (`gpflow`, `waveome.kernels`: the kernel classes of waveome_amd.kernels expose the
same attribute surface the stub touches - .kernels, .variance, .lengthscales, .period, .base_kernel, .active_dims,
Parameter.numpy() / .assign() / .trainable / .prior.scale).  CPU: it parses, and `encode` builds the flattened spec the
C-ABI accepts; GPU: `optimize_params_amd` fits three models and writes back the parameters Problem.fit finds."""
import ctypes as C
import os
import re
import sys
import types

import numpy as np
import pytest

from waveome_amd import ModelSpec, kernels as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"## The stub a maintainer would add.*?```python\n(.*?)```", text, re.S).group(1)
    lib = os.path.join(ROOT, "waveome_amd", "libwaveome_amd.so")
    assert 'C.CDLL("libwaveome_amd.so")' in code
    code = code.replace('C.CDLL("libwaveome_amd.so")', f"C.CDLL({lib!r})")
    gp = types.ModuleType("gpflow")
    gp.kernels = types.SimpleNamespace(
        SquaredExponential=K.SquaredExponential, Matern12=K.Matern12, Matern32=K.Matern32, Matern52=K.Matern52,
        Periodic=K.Periodic, Constant=K.Constant, Sum=K.Sum, Product=K.Product)
    wk = types.ModuleType("waveome.kernels")
    wk.Categorical, wk.Lin, wk.Poly, wk.Empty = K.Categorical, K.Lin, K.Poly, K.Empty
    w = types.ModuleType("waveome")
    w.kernels = wk
    saved = {k: sys.modules.get(k) for k in ("gpflow", "waveome", "waveome.kernels")}
    sys.modules.update({"gpflow": gp, "waveome": w, "waveome.kernels": wk})
    try:
        ns = {}
        exec(compile(code, "INTEGRATION.md:stub", "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ns


class _Model:   # what the stub reads off a waveome PSVGP
    def __init__(self, kernel):
        self.kernel = kernel
        self.likelihood = K.GaussianLikelihood()
        self.mean_function = K.ConstantMean()


def _kernel():
    frozen = K.Categorical(active_dims=[2])
    K.set_trainable(frozen.variance, False)
    k = K.Sum([K.SquaredExponential(active_dims=[0]), K.Periodic(K.SquaredExponential(active_dims=[1])),
               K.Product([frozen, K.Matern32(active_dims=[0])]), K.Lin(active_dims=[1])])
    k.kernels[0].variance.prior = K.Horseshoe(1.0)
    return k


def test_stub_encodes_the_spec_the_abi_accepts():
    ns = _stub_namespace()
    m = _Model(_kernel())
    spec, plist, keep = ns["encode"](m, 3)
    assert spec.n_terms == 4 and spec.n_dims == 3
    nt = C.c_int32()
    assert ns["_lib"].wv_spec_check(C.byref(spec), C.byref(nt)) == 0
    mine = ModelSpec(m.kernel, 3, m.likelihood, m.mean_function)
    assert nt.value == mine.n_trainable == len([p for p in plist if p.trainable])
    # same flattened structure as this repository's own encoder
    ours = mine.to_ctypes()
    assert [spec.factors[i].type for i in range(5)] == [ours.factors[i].type for i in range(5)]
    assert [spec.factors[i].dim_mask for i in range(5)] == [ours.factors[i].dim_mask for i in range(5)]
    assert spec.params[0].prior_scale == 1.0 and spec.params[1].prior_scale < 0


@pytest.mark.gpu
def test_stub_fits_and_writes_back_what_problem_fit_finds():
    from waveome_amd.engine import Context, Problem, lbfgs_opts

    ns = _stub_namespace()
    rng = np.random.default_rng(3)
    N = 120
    X = np.column_stack([rng.normal(size=N), rng.uniform(0, 4, N), rng.integers(0, 3, N).astype(float)])
    Y = np.stack([np.sin(X[:, 0] * (1 + p)) + 0.3 * np.cos(2 * X[:, 1]) + 0.2 * (X[:, 2] == p) + 0.1 * rng.normal(size=N)
                  for p in range(3)], axis=1)
    models = [_Model(_kernel()) for _ in range(3)]
    status = ns["optimize_params_amd"](models, X, Y, maxiter=200)
    ms = ModelSpec(_kernel(), 3, K.GaussianLikelihood(), K.ConstantMean())
    ctx = Context(0)
    prob = Problem(ctx, X, Y, ms, max_batch=3)
    ref = prob.fit(np.arange(3, dtype=np.int32), np.tile(ms.unconstrain(), (3, 1)), lbfgs_opts(maxiter=200, maxfun=200))
    prob.close()
    ctx.close()
    np.testing.assert_array_equal(status, ref["status"])
    for t, m in enumerate(models):
        got = ModelSpec(m.kernel, 3, m.likelihood, m.mean_function).theta()
        np.testing.assert_array_equal(got, ms.constrain(ref["u"][t]))
        assert m.kernel.kernels[2].kernels[0].variance.value == 1.0     # the frozen Categorical variance stayed
