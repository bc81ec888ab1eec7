"""float32 storage path (context dtype "float32": the reference's `set_precision("float32")`,
waveome/utilities.py:34-49; BASELINE config C5) against float64 references.

The N x N work matrices and their MFMA products are float32, everything else float64, so the
tolerances are float32 ones and are written next to each assertion."""
import os

import numpy as np
import pytest

from oracle import gp_oracle as o
from waveome_amd import ModelSpec
from waveome_amd import kernels as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx32():
    from waveome_amd.engine import Context

    c = Context(0, dtype="float32")
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx64():
    from waveome_amd.engine import Context

    c = Context(0)
    yield c
    c.close()


def _data(seed, N, D=3, P=2):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, D))
    Y = np.stack([np.sin((1 + p) * X[:, 0]) + 0.4 * X[:, 1] + 0.2 * rng.normal(size=N) for p in range(P)], axis=1)
    return X, Y


def _kernel():
    return K.Sum([K.SquaredExponential(active_dims=[0]), K.Matern32(active_dims=[1]),
                  K.Periodic(K.SquaredExponential(active_dims=[2]))])


@pytest.mark.parametrize("a_kc", [True, False])
@pytest.mark.parametrize("nb", [3, 9])  # 9: the batch -> XCD mapping of launches with >= 8 matrices
# partial row and column tiles (the operand loads redirect whole 32-row pieces past the extent: their products land in
# masked C elements only), one tile smaller than the tiling, and a contraction of 34 chunks (two float64 flushes + a rest)
@pytest.mark.parametrize("shape", [(256, 192, 128), (192, 320, 64), (64, 64, 64), (320, 64, 1088)])
def test_f32_mfma_gemm(ctx32, a_kc, nb, shape):
    rng = np.random.default_rng(1)
    M, N, Kd = shape
    A = rng.normal(size=(nb, M, Kd) if a_kc else (nb, Kd, M)).astype(np.float32)
    B = rng.normal(size=(nb, Kd, N)).astype(np.float32)
    C0 = rng.normal(size=(nb, M, N)).astype(np.float32)
    Am = A if a_kc else np.swapaxes(A, 1, 2)
    want = -0.5 * np.einsum("bmk,bkn->bmn", Am.astype(np.float64), B.astype(np.float64)) + 2.0 * C0
    got = ctx32.dbg_gemm(A, B, C0, a_kcontig=a_kc, b_kcontig=False, alpha=-0.5, beta=2.0)
    # float32 accumulation over <= 512 terms of size ~1 between float64 flushes: a few 1e-6 relative to sqrt(K)
    tol = 2e-4 * max(1.0, np.sqrt(Kd / 128.0))
    assert np.max(np.abs(got - want)) < tol
    got0 = ctx32.dbg_gemm(A, B, np.zeros_like(C0), a_kcontig=a_kc, b_kcontig=False)
    assert np.max(np.abs(got0 - np.einsum("bmk,bkn->bmn", Am.astype(np.float64), B.astype(np.float64)))) < tol


def test_f32_factor_inverse_stages(ctx32, ctx64):
    from waveome_amd.engine import Problem

    N = 300  # Npad = 320: uneven splits of the recursion
    X, Y = _data(5, N)
    spec = ModelSpec(_kernel(), 3, K.GaussianLikelihood(), K.ConstantMean())
    theta = spec.theta()
    p64 = Problem(ctx64, X, Y, spec, max_batch=1)
    Ky = p64.gram(theta)
    p64.close()
    p32 = Problem(ctx32, X, Y, spec, max_batch=1)
    K32 = p32.gram(theta)
    assert np.max(np.abs(K32 - Ky)) < 5e-7 * np.max(np.abs(Ky))  # one float32 rounding of each element
    L = np.linalg.cholesky(Ky)
    Minv = np.linalg.inv(L)
    Mgot = p32.dbg_stage(0, theta, 1)[:N, :N]
    assert np.max(np.abs(np.tril(Mgot) - Minv)) < 2e-4 * np.max(np.abs(Minv))
    Kinv = np.linalg.inv(Ky)
    Kgot = np.tril(p32.dbg_stage(0, theta, 2)[:N, :N])
    assert np.max(np.abs(Kgot - np.tril(Kinv))) < 2e-4 * np.max(np.abs(Kinv))
    p32.close()


@pytest.mark.parametrize("N", [64, 200, 1000])
def test_f32_value_and_gradient_close_to_float64(ctx32, ctx64, N):
    from waveome_amd.engine import Problem

    X, Y = _data(7, N)
    spec = ModelSpec(_kernel(), 3, K.GaussianLikelihood(), K.ConstantMean())
    rng = np.random.default_rng(2)
    U = spec.unconstrain()[None, :] + 0.3 * rng.normal(size=(4, spec.n_trainable))
    cols = np.array([0, 1, 0, 1], dtype=np.int32)
    p64 = Problem(ctx64, X, Y, spec, max_batch=4)
    f64, g64, _ = p64.value_grad(cols, U)
    l64, _ = p64.lml(cols, np.stack([spec.constrain(u) for u in U]))
    p64.close()
    p32 = Problem(ctx32, X, Y, spec, max_batch=4)
    f32, g32, st = p32.value_grad(cols, U)
    l32, _ = p32.lml(cols, np.stack([spec.constrain(u) for u in U]))
    p32.close()
    assert np.all(st == 0)
    # float32 operands, float64-flushed accumulation, value corrected to first order in E = M K_y M' - I (K_y in float64):
    # the north star's 1e-6 on the value; the gradient is that of the float32 factor (1e-6 ... 1e-5 of its largest entry)
    np.testing.assert_allclose(l32, l64, rtol=1e-6)
    np.testing.assert_allclose(f32, f64, rtol=1e-6)
    assert np.max(np.abs(g32 - g64)) < 1e-4 * np.max(np.abs(g64))


def test_f32_fit_reaches_the_float64_optimum(ctx32, ctx64):
    from waveome_amd.engine import Problem, lbfgs_opts

    X, Y = _data(9, 400)
    kern = K.Sum([K.SquaredExponential(active_dims=[0]), K.Lin(active_dims=[1])])
    spec = ModelSpec(kern, 3, K.GaussianLikelihood(), K.ConstantMean())
    U0 = np.tile(spec.unconstrain(), (2, 1))
    cols = np.array([0, 1], dtype=np.int32)
    p64 = Problem(ctx64, X, Y, spec, max_batch=2)
    r64 = p64.fit(cols, U0, lbfgs_opts())
    p64.close()
    p32 = Problem(ctx32, X, Y, spec, max_batch=2)
    r32 = p32.fit(cols, U0, lbfgs_opts())
    p32.close()
    # float32 noise in the objective stops L-BFGS-B a little earlier; the optimum is the same
    np.testing.assert_allclose(r32["loss"], r64["loss"], rtol=1e-4)
    th32 = np.stack([spec.constrain(u) for u in r32["u"]])
    th64 = np.stack([spec.constrain(u) for u in r64["u"]])
    # the constant mean is a nearly flat direction next to a long-lengthscale SE term: loose on it alone
    np.testing.assert_allclose(th32[:, :-1], th64[:, :-1], rtol=5e-2, atol=1e-2)
    np.testing.assert_allclose(th32[:, -1], th64[:, -1], atol=0.1)


def test_post_fit_calls_on_a_float32_problem_are_the_float64_ones(ctx32):
    """wv_predict / wv_whitened_means / wv_whitened_state are float64 computations whatever the problem's storage type:
    a float32 problem serves them on a float64 twin it builds from its own device copies (include/waveome_amd.h) -
    same numbers as the oracle, at float64 tolerances, column after column."""
    from oracle import gp_oracle as o
    from waveome_amd.engine import Problem

    X, Y = _data(3, 100)
    kern = K.Sum([K.SquaredExponential(active_dims=[0]), K.Lin(active_dims=[1])])
    spec = ModelSpec(kern, 3, K.GaussianLikelihood(), K.ConstantMean())
    d = spec.to_dict()
    theta = spec.constrain(spec.unconstrain() + 0.2)
    rng = np.random.default_rng(8)
    Xnew = np.column_stack([rng.uniform(-2, 2, 37), rng.normal(size=37), rng.integers(0, 3, 37).astype(float)])
    p = Problem(ctx32, X, Y, spec, max_batch=2)
    for col in (1, 0, 1):   # (a change of column re-uses the twin)
        m, v = p.predict(col, theta)
        mo, vo = o.predict_train(d, theta, X, Y[:, col])
        np.testing.assert_allclose(m, mo, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(v, vo, rtol=1e-8, atol=1e-10)
        m, v = p.predict(col, theta, include_noise=False, Xnew=Xnew)
        mo, vo = o.predict(d, theta, X, Y[:, col], Xnew, include_noise=False)
        np.testing.assert_allclose(m, mo, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(v, vo, rtol=1e-8, atol=1e-10)
        wm = p.whitened_means(col, theta, [0b01, 0b10, 0b11])
        wo = o.whitened_means(d, theta, X, Y[:, col], [[0], [1], [0, 1]])
        np.testing.assert_allclose(wm, wo, rtol=0, atol=2e-6)   # (chol(K + 1e-6 I): the reference's own reproducibility)
    q_mu, q_sqrt = p.whitened_state(0, theta)
    assert q_mu.shape == (100,) and np.all(np.isfinite(q_sqrt)) and np.allclose(np.triu(q_sqrt, 1), 0.0)
    # the float32 fits of the same problem are untouched by the twin
    f, _, st = p.value_grad(np.array([0], dtype=np.int32), spec.unconstrain()[None, :])
    assert st[0] == 0 and np.isfinite(f[0])
    p.close()


def test_set_precision_switches_the_default_context():
    from waveome_amd import utilities as U
    from waveome_amd.model_classes import default_context

    with pytest.raises(ValueError):
        U.set_precision("float16")
    import warnings

    with warnings.catch_warnings():   # a parity mode since round 5: nothing to warn about
        warnings.simplefilter("error")
        U.set_precision("float32")
    try:
        assert U.get_precision() == "float32" and default_context().dtype == "float32"
    finally:
        U.set_precision("float64")
    assert default_context().dtype == "float64"


def test_config_c5_float32_n16384(ctx32, ctx64):
    """BASELINE config C5: N = 16384, D = 4, sum_d Matern32[d], float32; value-only and value+gradient
    against the float64 path on the same inputs."""
    import time

    from waveome_amd.engine import Problem

    N, D = 16384, 4
    rng = np.random.default_rng(9102 + 5)
    X = rng.uniform(-3, 3, size=(N, D))
    X = (X - X.mean(0)) / X.std(0, ddof=1)
    Y = (np.sin(1.3 * X[:, 0]) + 0.5 * np.sin(2.1 * X[:, 1]) + 0.1 * rng.normal(size=N))[:, None]
    spec = ModelSpec(K.Sum([K.Matern32(active_dims=[d]) for d in range(D)]), D, K.GaussianLikelihood(),
                     K.ConstantMean())
    th = spec.theta()
    th[spec.noise] = 0.3
    u = spec.unconstrain(th)[None, :]
    cols = np.zeros(1, dtype=np.int32)
    p64 = Problem(ctx64, X, Y, spec, max_batch=1)
    f64, g64, _ = p64.value_grad(cols, u)
    p64.close()
    p32 = Problem(ctx32, X, Y, spec, max_batch=1)
    p32.value_grad(cols, u)  # warm-up
    t0 = time.perf_counter()
    l32, st = p32.lml(cols, th[None, :])
    t1 = time.perf_counter()
    f32, g32, st2 = p32.value_grad(cols, u)
    t2 = time.perf_counter()
    p32.close()
    assert st[0] == 0 and st2[0] == 0
    print(f"C5 float32: value-only {t1 - t0:.3f} s ({N ** 3 / 3 / (t1 - t0) / 1e12:.1f} TFLOP/s), "
          f"value+grad {t2 - t1:.3f} s ({N ** 3 / (t2 - t1) / 1e12:.1f} TFLOP/s); "
          f"rel err value {abs(f32[0] - f64[0]) / abs(f64[0]):.2e}, "
          f"grad {np.max(np.abs(g32 - g64)) / np.max(np.abs(g64)):.2e}")
    # float32 as a PARITY mode - the north star's 1e-6 on the value (measured 1.4e-7), 1e-4 of the
    # largest component on the gradient (measured 1e-6)
    assert abs(f32[0] - f64[0]) <= 1e-6 * abs(f64[0])
    assert abs(-l32[0] - f64[0]) <= 1e-6 * abs(f64[0])
    assert np.max(np.abs(g32 - g64)) <= 1e-4 * np.max(np.abs(g64))


def test_f32_matern32_against_the_state_space_likelihood_n8192(ctx32, ctx64):
    """C5's kernel (Matern-3/2) on the float32 path against an answer that is neither the float64 device path nor the
    oracle: the Kalman-filter likelihood of its stochastic differential equation (tests/statespace_matern.py), N = 8192
    on one column.  The float64 path lands on it to 1e-8; the float32 path, with its value corrected back through
    E = M K_y M' - I, to the north star's 1e-6."""
    import statespace_matern as S

    from waveome_amd.engine import Problem

    rng = np.random.default_rng(6)
    N = 8192
    x = rng.uniform(0, 80, N)
    y = np.sin(x) + 0.4 * np.cos(0.37 * x) + 0.3 * rng.normal(size=N)
    var, ell, s2, c = 1.4, 0.8, 0.3, 0.1
    ms = ModelSpec(K.Matern32(variance=var, lengthscales=ell, active_dims=[0]), 1, K.GaussianLikelihood(variance=s2),
                   K.ConstantMean(c))
    want = S.kalman_lml("m32", var, ell, s2, x, y, mean=c)
    got = {}
    for name, cx in (("f64", ctx64), ("f32", ctx32)):
        prob = Problem(cx, x[:, None], y[:, None], ms, max_batch=1)
        lml, st = prob.lml([0], ms.theta()[None, :])
        f, _, st2 = prob.value_grad([0], ms.unconstrain()[None, :])
        prob.close()
        assert st[0] == 0 and st2[0] == 0
        got[name] = (lml[0], -f[0])
    assert got["f64"][0] == pytest.approx(want, rel=1e-8) and got["f64"][1] == pytest.approx(want, rel=1e-8)
    print(f"float32 vs the state-space likelihood at N = {N}: value-only {abs(got['f32'][0] - want) / abs(want):.1e}, "
          f"value+gradient {abs(got['f32'][1] - want) / abs(want):.1e}")
    assert got["f32"][0] == pytest.approx(want, rel=1e-6) and got["f32"][1] == pytest.approx(want, rel=1e-6)


@pytest.mark.parametrize("N", [1, 2, 63, 65, 129])
def test_f32_ragged_and_tiny_sizes(ctx32, ctx64, N):
    from waveome_amd.engine import Problem

    X, Y = _data(11, N)
    spec = ModelSpec(K.Sum([K.SquaredExponential(active_dims=[0]), K.Lin(active_dims=[1])]), 3,
                     K.GaussianLikelihood(), K.ConstantMean())
    u = spec.unconstrain()[None, :]
    cols = np.zeros(1, dtype=np.int32)
    p64 = Problem(ctx64, X, Y, spec, max_batch=1)
    f64, g64, _ = p64.value_grad(cols, u)
    p64.close()
    p32 = Problem(ctx32, X, Y, spec, max_batch=1)
    f32, g32, st = p32.value_grad(cols, u)
    p32.close()
    assert st[0] == 0
    assert abs(f32[0] - f64[0]) <= 1e-6 * max(1.0, abs(f64[0]))
    assert np.max(np.abs(g32 - g64)) <= 1e-4 * max(1.0, np.max(np.abs(g64)))


def test_f32_multi_structure_fit(ctx32, ctx64):
    """The multi-structure work queue on a float32 context: same optima as on the float64 one."""
    from waveome_amd.engine import Problem, lbfgs_opts

    X, Y = _data(13, 300)
    specs = [ModelSpec(k, 3, K.GaussianLikelihood(), K.ConstantMean()) for k in
             (K.SquaredExponential(active_dims=[0]), K.Sum([K.Matern32(active_dims=[0]), K.Lin(active_dims=[1])]))]
    cols = np.array([0, 1, 1, 0], dtype=np.int32)
    sidx = np.array([0, 1, 0, 1], dtype=np.int32)
    width = max(sp.n_trainable for sp in specs)
    U0 = np.zeros((4, width))
    for t, si in enumerate(sidx):
        U0[t, :specs[si].n_trainable] = specs[si].unconstrain()
    out = {}
    for name, c in (("f32", ctx32), ("f64", ctx64)):
        prob = Problem(c, X, Y, specs, max_batch=4)
        out[name] = prob.fit_multi(cols, sidx, U0, lbfgs_opts())
        prob.close()
    np.testing.assert_allclose(out["f32"]["loss"], out["f64"]["loss"], rtol=1e-4)


def test_penalized_optimization_end_to_end_in_float32():
    """set_precision("float32") (waveome/utilities.py:34-49) then the whole GPSearch.penalized_optimization():
    fits run on the float32 path, the float64-only post-fit steps on a float64 context of the same device."""
    import pandas as pd

    from waveome_amd import utilities as U
    from waveome_amd.model_search import GPSearch

    rng = np.random.default_rng(4)
    n = 120
    X = pd.DataFrame({"t": rng.uniform(0, 5, n), "dose": rng.normal(size=n), "grp": rng.integers(0, 3, n).astype(float)})
    Y = pd.DataFrame({"a": np.sin(2 * X["t"]) + 0.2 * rng.normal(size=n), "b": 0.8 * X["dose"] + 0.2 * rng.normal(size=n)})
    out = {}
    for prec in ("float64", "float32"):
        U.set_precision(prec)
        try:
            gps = GPSearch(X, Y, categorical_vars=["grp"], Y_transform="standardize")
            gps.penalized_optimization(penalization_factor=10.0)
            out[prec] = {f: (m.fit_result["loss"], m.kernel_name, list(m.feature_importances)) for f, m in gps.models.items()}
        finally:
            U.set_precision("float64")
    for f in ("a", "b"):
        # float32 rounding noise can stop L-BFGS-B a little earlier on the penalised, partly flat objective:
        # the optimum is the same to a few per cent, and the terms that survive pruning largely coincide
        assert abs(out["float32"][f][0] - out["float64"][f][0]) <= 0.05 * max(1.0, abs(out["float64"][f][0]))
        n32, n64 = out["float32"][f][1], out["float64"][f][1]
        n32, n64 = ([n32] if isinstance(n32, str) else list(n32)), ([n64] if isinstance(n64, str) else list(n64))
        assert set(n32) & set(n64)
    # the iterated analytic factor (predict_y on a float64 twin) also runs through on a float32 context
    U.set_precision("float32")
    try:
        gps = GPSearch(X, Y, categorical_vars=["grp"], Y_transform="standardize")
        gps.penalized_optimization(penalization_factor=None, num_factor_iter=2)
        assert all(np.isfinite(m.fit_result["loss"]) and m.penalization_factor > 0 for m in gps.models.values())
    finally:
        U.set_precision("float64")


_ORDER_SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem
N, D = 4096, 4
rng = np.random.default_rng(9102 + 5)
X = rng.uniform(-3, 3, size=(N, D)); X = (X - X.mean(0)) / X.std(0, ddof=1)
Y = (np.sin(X[:, 0]) + 0.5 * np.sin(2.1 * X[:, 1]) + 0.1 * rng.normal(size=N))[:, None]
kern = K.Sum([K.SquaredExponential(active_dims=[0]), K.Periodic(K.SquaredExponential(active_dims=[1])), K.SquaredExponential(active_dims=[2])])
spec = ModelSpec(kern, D, K.GaussianLikelihood(), K.ConstantMean())
th = spec.theta(); th[spec.noise] = 0.01        # cond(K_y) ~ 8e5
out = []
for dt in ("float64", "float32"):
    p = Problem(Context(0, dtype=dt), X, Y, spec, max_batch=1)
    l, st = p.lml([0], th[None, :]); f, g, st2 = p.value_grad([0], spec.unconstrain(th)[None, :])
    assert st[0] == 0 and st2[0] == 0
    out += [l[0], f[0]]
print(*out)
"""


def test_f32_higher_orders_for_an_ill_conditioned_gram_matrix(tmp_path):
    """cond(K_y) ~ 1e6 (noise variance 0.01): E = M K_y M' - I of the float32 factor is no longer tiny, the first-order
    value correction leaves ~1e-5 relative, and WV_F32_ORDER=4 (second to fourth order: |E|_F^2 by a fixed Rademacher probe,
    the quadratic form's terms exactly - two more applications of E) takes it below 1e-6.  Read once per process: child
    processes."""
    import subprocess
    import sys

    root = str(__import__("pathlib").Path(__file__).resolve().parents[1])
    script = tmp_path / "order.py"
    script.write_text(_ORDER_SCRIPT)
    err = {}
    for order in ("1", "2", "4"):
        r = subprocess.run([sys.executable, str(script), root], env=dict(os.environ, WV_F32_ORDER=order), capture_output=True,
                           timeout=600, check=True)
        l64, f64, l32, f32 = (float(v) for v in r.stdout.decode().split())
        assert l64 == pytest.approx(-f64, rel=1e-10)
        err[order] = max(abs(l32 - l64), abs(f32 - f64)) / abs(l64)
    print("float32 value error by WV_F32_ORDER at cond ~ 8e5:", err)
    assert err["1"] < 1e-4 and err["2"] < 1e-6 and err["4"] < 1e-6 and max(err["2"], err["4"]) < 0.1 * err["1"]


@pytest.mark.parametrize("seed", range(40, 52))
def test_f32_random_kernel_trees_match_the_float64_oracle(ctx32, seed):
    """The float32 context on seeded random Sum / Product trees over every kernel type (the generator of
    tests/test_gpu_parity.py): frozen variances, Polynomial and Empty factors and products among them - the terms whose
    K_y-weighted sum the trace pass cannot take from a gradient sum and evaluates explicitly (trace_kernel, KSUM).  Loss / LML
    1e-6, gradient 1e-4 against the float64 oracle."""
    import test_gpu_parity as T

    from waveome_amd.engine import Problem

    rng = np.random.default_rng(seed)
    kern = T._random_tree(rng, int(rng.integers(1, 4)))
    ms = ModelSpec(kern, 4, K.GaussianLikelihood(), K.ConstantMean() if rng.random() < 0.7 else None)
    if rng.random() < 0.6:
        T._with_priors(ms, scale=float(rng.choice([0.1, 1.0, 10.0])))
    if sum(len(t) for t in ms.terms) > 128 or len(ms.terms) > 64 or max(len(t) for t in ms.terms) > 4:
        pytest.skip("beyond the limits of one device spec")
    N = int(rng.choice([17, 64, 129, 200, 333, 400]))
    X, Y = T._data(seed, N)
    nb = int(rng.integers(1, 4))
    prob = Problem(ctx32, X, Y, ms, max_batch=nb)
    d = ms.to_dict()
    U = np.stack([ms.unconstrain() + 0.3 * rng.normal(size=ms.n_trainable) for _ in range(nb)])
    cols = rng.integers(0, Y.shape[1], nb).astype(np.int32)
    loss, grad, st = prob.value_grad(cols, U)
    lml, st2 = prob.lml(cols, np.stack([ms.constrain(u) for u in U]))
    prob.close()
    for b in range(nb):
        fo, go = o.objective(d, U[b], X, Y[:, cols[b]])
        if not np.isfinite(fo):
            continue
        assert st[b] == 0 and st2[b] == 0
        assert abs(loss[b] - fo) <= 1e-6 * max(1.0, abs(fo)), (seed, b, loss[b], fo)
        assert abs(lml[b] - o.lml(d, ms.constrain(U[b]), X, Y[:, cols[b]])) <= 1e-6 * max(1.0, abs(fo))
        assert np.max(np.abs(grad[b] - go)) <= 1e-4 * max(1.0, np.max(np.abs(go))), (seed, b)
