"""Missing output values (`wv_problem_caps::nan_is_missing`): a NaN in Y[:, p] removes that row from output p's
fits - the complete-case filter the reference applies per output in its search (waveome/model_search.py:3034-3043) -
while all outputs stay in one batch.  Checked against the oracle run on the filtered rows."""
import numpy as np
import pytest

from oracle import gp_oracle as o
from waveome_amd import ModelSpec
from waveome_amd import kernels as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from waveome_amd.engine import Context

    c = Context(0)
    yield c
    c.close()


def _data(seed, N, P=4, frac=0.2):
    rng = np.random.default_rng(seed)
    X = np.column_stack([rng.uniform(-2, 2, N), rng.normal(size=N), rng.integers(0, 4, N).astype(float)])
    Y = np.stack([np.sin((1 + 0.4 * p) * X[:, 0]) + 0.3 * X[:, 1] + 0.2 * (X[:, 2] == p % 4) + 0.15 * rng.normal(size=N)
                  for p in range(P)], axis=1)
    miss = rng.random(size=Y.shape) < frac
    miss[:, 0] = False          # one complete column
    miss[N // 2:, 1] = True     # one column with its whole second half missing (crosses 64-row granules)
    Y[miss] = np.nan
    return X, Y


def _kernel():
    return K.Sum([K.SquaredExponential(active_dims=[0]), K.Lin(active_dims=[1]), K.Categorical(active_dims=[2]),
                  K.Product([K.Categorical(active_dims=[2]), K.Matern32(active_dims=[0])])])


@pytest.mark.parametrize("N", [150, 333])
def test_value_gradient_and_lml_are_those_of_the_complete_cases(ctx, N):
    from waveome_amd.engine import Problem

    X, Y = _data(1, N)
    spec = ModelSpec(_kernel(), 3, K.GaussianLikelihood(), K.ConstantMean())
    d = spec.to_dict()
    rng = np.random.default_rng(2)
    cols = np.array([0, 1, 2, 3, 1, 3], dtype=np.int32)
    U = spec.unconstrain()[None, :] + 0.3 * rng.normal(size=(len(cols), spec.n_trainable))
    p = Problem(ctx, X, Y, spec, max_batch=len(cols), nan_is_missing=True)
    f, g, st = p.value_grad(cols, U)
    thetas = np.stack([spec.constrain(u) for u in U])
    lml, st2 = p.lml(cols, thetas)
    assert np.all(st == 0) and np.all(st2 == 0)
    for b, c in enumerate(cols):
        seen = ~np.isnan(Y[:, c])
        fo, go = o.objective(d, U[b], X[seen], Y[seen, c])
        assert f[b] == pytest.approx(fo, rel=1e-10, abs=1e-10)
        assert np.max(np.abs(g[b] - go)) <= 1e-7 * max(1.0, np.max(np.abs(go)))
        assert lml[b] == pytest.approx(o.lml(d, thetas[b], X[seen], Y[seen, c]), rel=1e-10, abs=1e-10)
    # post-fit calls: those of the complete cases of the column.  Training inputs answer all N rows - a row missing for
    # the column is predicted like any new input; the component means have no value there (NaN)
    rng = np.random.default_rng(3)
    Xnew = np.column_stack([rng.uniform(-2, 2, 29), rng.normal(size=29), rng.integers(0, 4, 29).astype(float)])
    for c in (1, 0, 3):
        seen = ~np.isnan(Y[:, c])
        th = thetas[c]
        m, v = p.predict(c, th)
        mo, vo = o.predict_train(d, th, X[seen], Y[seen, c])
        np.testing.assert_allclose(m[seen], mo, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(v[seen], vo, rtol=1e-8, atol=1e-10)
        if (~seen).any():
            mo, vo = o.predict(d, th, X[seen], Y[seen, c], X[~seen])
            np.testing.assert_allclose(m[~seen], mo, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(v[~seen], vo, rtol=1e-8, atol=1e-10)
        m, v = p.predict(c, th, include_noise=False, Xnew=Xnew)
        mo, vo = o.predict(d, th, X[seen], Y[seen, c], Xnew, include_noise=False)
        np.testing.assert_allclose(m, mo, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(v, vo, rtol=1e-8, atol=1e-10)
        wm = p.whitened_means(c, th, [0b0001, 0b1110, 0b1111])
        wo = o.whitened_means(d, th, X[seen], Y[seen, c], [[0], [1, 2, 3], [0, 1, 2, 3]])
        np.testing.assert_allclose(wm[:, seen], wo, rtol=0, atol=2e-6)   # (chol(K + 1e-6 I): the reference's own reproducibility)
        assert np.all(np.isnan(wm[:, ~seen]))
    with pytest.raises(Exception, match="missing outputs"):
        p.whitened_state(1, thetas[1])
    # the fits of the problem are untouched by the twin
    f3, _, _ = p.value_grad(cols, U)
    np.testing.assert_array_equal(f3, f)
    p.close()
    # without the flag NaN stays data: every evaluation of such a column is non-finite, the complete column is fine
    q = Problem(ctx, X, Y, spec, max_batch=2)
    f2, _, st3 = q.value_grad(np.array([0, 1], dtype=np.int32), U[:2])
    assert f2[0] == pytest.approx(f[0], rel=1e-12) and (not np.isfinite(f2[1]) or st3[1] != 0)
    q.close()


def test_fits_with_missing_outputs_match_fits_on_filtered_data(ctx):
    from waveome_amd.engine import Problem, lbfgs_opts

    X, Y = _data(4, 200)
    spec = ModelSpec(K.Sum([K.SquaredExponential(active_dims=[0]), K.Lin(active_dims=[1])]), 3,
                     K.GaussianLikelihood(), K.ConstantMean())
    opts = lbfgs_opts(maxiter=60)
    cols = np.arange(4, dtype=np.int32)
    U0 = np.tile(spec.unconstrain(), (4, 1))
    p = Problem(ctx, X, Y, spec, max_batch=4, nan_is_missing=True)
    r = p.fit(cols, U0, opts)
    p.close()
    for c in cols:
        seen = ~np.isnan(Y[:, c])
        q = Problem(ctx, X[seen], Y[seen, c:c + 1], spec, max_batch=1)
        rq = q.fit(np.zeros(1, dtype=np.int32), U0[:1], opts)
        q.close()
        # same objective, another summation order (padding rows sit elsewhere): the trajectories agree to rounding
        assert r["loss"][c] == pytest.approx(rq["loss"][0], rel=1e-7, abs=1e-7)
        np.testing.assert_allclose(spec.constrain(r["u"][c]), spec.constrain(rq["u"][0]), rtol=1e-4, atol=1e-5)


def test_search_with_missing_values_equals_the_oracle_scored_search(ctx):
    """The kernel search on data with missing inputs and outputs (direct callers of the search, as with the
    reference's full_kernel_search; GPSearch refuses NaN like the reference's): rows with a missing input go for every
    output, a missing output value only for that output - the same search tree as the CPU oracle scoring each
    output on its complete cases."""
    from tests.test_search import OracleScorer
    from waveome_amd import kernel_search as ks

    rng = np.random.default_rng(7)
    n = 60
    X = np.column_stack([rng.normal(size=n), rng.normal(size=n), (np.arange(n) % 3).astype(float)])
    Y = np.column_stack([1.5 * X[:, 1] + 0.2 * rng.normal(size=n), np.sin(2.0 * X[:, 0]) + 0.15 * rng.normal(size=n)])
    X[[3, 17], 0] = np.nan
    Y[rng.choice(n, 12, replace=False), 0] = np.nan
    Y[rng.choice(n, 9, replace=False), 1] = np.nan
    kl = [K.SquaredExponential(), K.Lin()]
    scorer = ks.GpuScorer(X, Y, context=ctx)
    assert scorer.nan_is_missing and scorer.X.shape[0] == n - 2
    got = ks.run_searches(range(2), 3, kl, scorer, cat_vars=[2], max_depth=2, keep_only_best=False)

    class CompleteCaseOracle(OracleScorer):
        def __call__(self, tasks):
            out = []
            for col, kern in tasks:
                seen = ~np.isnan(self.Y[:, col])
                sub = OracleScorer(self.X[seen], self.Y[seen])
                sub.cache = self.cache.setdefault(col, {})
                out.extend(sub([(col, kern)]))
            return out

    x_ok = ~np.isnan(X).any(axis=1)
    ref = ks.run_searches(range(2), 3, kl, CompleteCaseOracle(X[x_ok], Y[x_ok]), cat_vars=[2], max_depth=2,
                          keep_only_best=False)
    for p in range(2):
        assert set(got[p]["models"]) == set(ref[p]["models"])
        assert got[p]["best_model"] == ref[p]["best_model"]
        for name, m in got[p]["models"].items():
            assert m["bic"] == pytest.approx(ref[p]["models"][name]["bic"], abs=0.05), (p, name)
