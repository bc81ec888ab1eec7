"""Compositional kernel search: candidate generation and control flow on the CPU,
and the GPU-scored search against the same search scored by the CPU oracle.

The reference holds no tests for `run_search` / `full_kernel_search`; the expected candidate names below
are derived by hand from waveome/model_search.py:2347-2664 (naming `k.name + str(k.active_dims)`, the
`base_name < k_info` ordering rule, the categorical / two-way-interaction skips)."""
import copy
import sys
from pathlib import Path

import os

import numpy as np
import pandas as pd
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import gp_oracle as O  # noqa: E402
from waveome_amd import kernel_search as ks  # noqa: E402
from waveome_amd import kernels as K  # noqa: E402
from waveome_amd.spec import ModelSpec  # noqa: E402
from waveome_amd.utilities import check_if_model_exists  # noqa: E402


def default_kernels():
    return [K.SquaredExponential(), K.Matern12(), K.Lin(), K.Periodic(K.SquaredExponential())]


def test_check_if_model_exists_ignores_term_and_factor_order():
    models = ["lin[1]+squared_exponential[0]", "categorical[2]*lin[0]"]
    assert check_if_model_exists("squared_exponential[0]+lin[1]", models)
    assert check_if_model_exists("lin[0]*categorical[2]", models)
    assert not check_if_model_exists("lin[1]", models)
    assert not check_if_model_exists("lin[1]+squared_exponential[1]", models)


def test_depth1_candidates():
    c = ks.loc_kernel_candidates(3, default_kernels(), cat_vars=[2], depth=1)
    assert list(c) == ["squared_exponential[0]", "matern12[0]", "lin[0]", "periodic[0]", "constant",
                       "squared_exponential[1]", "matern12[1]", "lin[1]", "periodic[1]", "categorical[2]"]
    const = c["constant"]
    assert const.variance.value == 1e-6 and not const.variance.trainable
    assert c["periodic[1]"].active_dims == [1] and c["lin[1]"].active_dims == [1]
    # the caller's kernel list is never touched
    kl = default_kernels()
    ks.loc_kernel_candidates(3, kl, cat_vars=[2], depth=1)
    assert all(k.active_dims in (None, [0]) for k in kl)


def test_sum_and_product_candidates():
    kl = default_kernels()
    d1 = ks.loc_kernel_candidates(3, kl, cat_vars=[2], depth=1)
    base = copy.deepcopy(d1["squared_exponential[0]"])
    base.variance.assign(3.0)
    base.lengthscales.assign(0.2)
    s = ks.loc_kernel_candidates(3, kl, base_kern=base, base_name="squared_exponential[0]", cat_vars=[2], depth=2,
                                 operation="sum", prev_models=list(d1))
    assert list(s) == ["matern12[0]+squared_exponential[0]", "lin[0]+squared_exponential[0]",
                       "periodic[0]+squared_exponential[0]", "squared_exponential[0]+squared_exponential[1]",
                       "matern12[1]+squared_exponential[0]", "lin[1]+squared_exponential[0]",
                       "periodic[1]+squared_exponential[0]", "categorical[2]+squared_exponential[0]"]
    # the base restarts from 1.0 (model_search.py:2420-2421) and kernel order follows the name
    cand = s["lin[1]+squared_exponential[0]"]
    assert [k.name for k in cand.kernels] == ["lin", "squared_exponential"]
    assert cand.kernels[1].variance.value == 1.0 and cand.kernels[1].lengthscales.value == 1.0
    p = ks.loc_kernel_candidates(3, kl, base_kern=base, base_name="squared_exponential[0]", cat_vars=[2], depth=2,
                                 operation="product", prev_models=list(d1) + list(s))
    assert "categorical[2]*squared_exponential[0]" in p and "lin[1]*squared_exponential[0]" in p
    # the new factor's variance is frozen at 1 (model_search.py:2469-2472), periodic through its base kernel
    prod = p["periodic[1]*squared_exponential[0]"]
    assert not prod.kernels[0].base_kernel.variance.trainable and prod.kernels[1].variance.trainable
    # a product is not multiplied again, a categorical feature already in the base is skipped
    assert ks.loc_kernel_candidates(3, kl, base_kern=prod, base_name="periodic[1]*squared_exponential[0]",
                                    cat_vars=[2], depth=3, operation="product") == {}
    cat = d1["categorical[2]"]
    s2 = ks.loc_kernel_candidates(3, kl, base_kern=cat, base_name="categorical[2]", cat_vars=[2], depth=2,
                                  operation="sum", prev_models=list(d1))
    assert all("categorical[2]+" in n for n in s2) and len(s2) == 8
    # models fitted before are not generated again
    again = ks.loc_kernel_candidates(3, kl, base_kern=base, base_name="squared_exponential[0]", cat_vars=[2],
                                     depth=2, operation="sum", prev_models=list(d1) + list(s))
    assert again == {}


def test_split_product_candidates_keep_names_and_kernels_aligned():
    kl = [K.SquaredExponential(), K.Lin()]
    base = K.Sum([K.Categorical(active_dims=[2]), K.SquaredExponential(active_dims=[0])])
    c = ks.loc_kernel_candidates(3, kl, base_kern=base, base_name="categorical[2]+squared_exponential[0]",
                                 cat_vars=[2], depth=3, operation="split_product", prev_models=[])
    assert "categorical[2]*lin[1]+squared_exponential[0]" in c
    assert "categorical[2]+lin[1]*squared_exponential[0]" in c
    assert "categorical[2]+categorical[2]*squared_exponential[0]" in c
    assert not any(n.startswith("categorical[2]*categorical[2]") for n in c)
    for name, kern in c.items():
        parts = name.split("+")
        assert len(parts) == len(kern.kernels)
        for part, comp in zip(parts, kern.kernels):
            got = "*".join(ks.kernel_info(f) for f in comp.kernels) if comp.name == "product" else ks.kernel_info(comp)
            assert got == part


def test_prune_candidates():
    k = K.Sum([K.Product([K.Categorical(active_dims=[2]), K.Lin(active_dims=[1])]),
               K.SquaredExponential(active_dims=[0])])
    k.kernels[1].lengthscales.assign(0.3)
    c = ks.prune_candidates("categorical[2]*lin[1]+squared_exponential[0]", k, existing=["lin[1]"])
    assert set(c) == {"categorical[2]+squared_exponential[0]", "lin[1]+squared_exponential[0]",
                      "categorical[2]*lin[1]"}
    assert c["lin[1]+squared_exponential[0]"].kernels[1].lengthscales.value == 1.0  # reset (:2857-2858)
    # nothing to prune in a single base kernel; a lone product is split into its factors
    assert ks.prune_candidates("lin[1]", K.Lin(active_dims=[1]), []) == {}
    lone = ks.prune_candidates("categorical[2]*lin[1]", k.kernels[0], existing=["lin[1]"])
    assert list(lone) == ["categorical[2]"]
    # where a product's factor lands among the remaining terms: behind them when their name sorts first or EQUALS the
    # factor's label (the argsort of the two names at model_search.py:2933-2934), in front otherwise
    k2 = K.Sum([K.Lin(active_dims=[2]), K.Product([K.Periodic(K.SquaredExponential(active_dims=[0])), K.Lin(active_dims=[2])])])
    c2 = ks.prune_candidates("lin[2]+periodic[0]*lin[2]", k2, existing=[])
    assert list(c2) == ["periodic[0]*lin[2]", "lin[2]+periodic[0]", "lin[2]+lin[2]"]
    assert len(c2["lin[2]+lin[2]"].kernels) == 2


def _drive(table, **kw):
    """Run one search generator against a table name -> bic (missing names: 1000)."""
    g = ks.full_kernel_search_steps(2, [K.SquaredExponential(), K.Lin()], **kw)
    asked = []
    try:
        req = next(g)
        while True:
            asked.append(list(req))
            req = g.send({n: {"bic": table.get(n, 1000.0)} for n in req})
    except StopIteration as e:
        return e.value, asked


def test_search_stops_on_constant_and_on_no_improvement():
    res, asked = _drive({"constant": 1.0})
    assert res["best_model"] == "constant" and len(asked) == 1
    # depth 2 does not beat depth 1 -> stop after (pruning) depth 2
    res, asked = _drive({"lin[0]": 10.0, "lin[0]+squared_exponential[1]": 11.0})
    assert res["best_model"] == "lin[0]" and res["depth_reached"] == 2
    # keep_top_k: only candidates within metric_diff of the level's best are expanded
    res, asked = _drive({"lin[0]": 10.0, "lin[1]": 12.0, "squared_exponential[0]": 30.0,
                         "lin[0]+lin[1]": 5.0}, max_depth=2)
    assert res["best_model"] == "lin[0]+lin[1]"
    depth2 = asked[1]
    assert any(n.startswith("lin[0]+") or n.endswith("+lin[0]") for n in depth2)
    assert not any("squared_exponential[0]*" in n and "lin" not in n for n in depth2)
    assert all(("lin[0]" in n) or ("lin[1]" in n) for n in depth2)


def test_search_prune_inserts_only_improvements():
    table = {"lin[0]": 10.0, "lin[0]+squared_exponential[1]": 8.0,
             "lin[0]*squared_exponential[0]+squared_exponential[1]": 7.0,
             # pruning the product at depth 3 offers lin[0]+squared_exponential[1] (exists),
             # squared_exponential[0]+squared_exponential[1] and lin[0]*squared_exponential[0]
             "squared_exponential[0]+squared_exponential[1]": 6.5}
    res, asked = _drive(table, max_depth=3, keep_only_best=False)
    assert res["best_model"] == "squared_exponential[0]+squared_exponential[1]"
    e = res["models"]["squared_exponential[0]+squared_exponential[1]"]
    assert e["parent"] == "lin[0]*squared_exponential[0]+squared_exponential[1]" and e["depth"] == 3
    # pruned structures that do not beat the level's best are not recorded
    kids = [v for v in res["models"].values() if v["parent"] == "lin[0]*squared_exponential[0]+squared_exponential[1]"]
    assert len(kids) == 1


@pytest.fixture(autouse=True, scope="module")
def _one_blas_thread():
    """The oracle fits of this file are on 48-row matrices, where threaded OpenBLAS buys nothing - and in the 8-CPU build
    container a threaded LAPACK call in them (dpotri, 48 x 48) was seen to spin for minutes once in a few runs of the whole
    CPU suite (always in this file, which runs after the tests that start gloo rank processes).  One BLAS thread here
    makes the suite's time deterministic; the other files keep the thread count their chaotic-path tolerances were
    written with."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        yield
        return
    with threadpool_limits(limits=1):
        yield


class OracleScorer:
    """CPU stand-in for kernel_search.GpuScorer: the same scoring rule on SciPy fits of the oracle."""

    def __init__(self, X, Y):
        self.X, self.Y, self.cache = X, Y, {}
        self.n_fits = 0

    def __call__(self, tasks):
        out = []
        for col, kern in tasks:
            spec = ModelSpec(kern, self.X.shape[1], K.GaussianLikelihood(), K.ConstantMean())
            key = (col, ks.structure_signature(spec))
            if key not in self.cache:
                r = O.fit(spec.to_dict(), self.X, self.Y[:, col])
                k_count = sum(1 for q in spec.param_objs if q.trainable) + 2
                self.cache[key] = {"bic": round(2 * k_count + 2 * r["loss"], 2), "loss": r["loss"], "lml": r["lml"],
                                   "nit": r["nit"], "nfev": r["nfev"], "status": r["status"], "u": r["u"]}
                self.n_fits += 1
            out.append(self.cache[key])
        return out


def search_data(n=48, seed=3):
    rng = np.random.default_rng(seed)
    x0 = rng.normal(size=n)
    x1 = rng.normal(size=n)
    g = (np.arange(n) % 3).astype(float)
    X = pd.DataFrame({"t": x0, "dose": x1, "grp": g})
    Y = pd.DataFrame({"lin_dose": 1.5 * x1 + 0.2 * rng.normal(size=n),
                      "smooth_t": np.sin(2.0 * x0) + 0.15 * rng.normal(size=n),
                      "noise": rng.normal(size=n)})
    return X, Y


def test_oracle_scored_search_recovers_structure():
    from waveome_amd.model_search import GPSearch

    X, Y = search_data()
    gps = GPSearch(X, Y, categorical_vars=["grp"])
    Xn, Yn = gps.X.to_numpy(), gps.Y.to_numpy()
    res = ks.run_searches(range(3), 3, [K.SquaredExponential(), K.Lin()], OracleScorer(Xn, Yn), cat_vars=[2],
                          max_depth=2)
    assert "lin[1]" in res[0]["best_model"]
    assert "[0]" in res[1]["best_model"] and "lin[1]" not in res[1]["best_model"]
    assert res[2]["best_model"] == "constant"


class _DelayedOracleSession:
    """Stand-in for kernel_search._SearchSession on the CPU: fits come from the oracle scorer, but a request is
    answered only a few `pump` calls later and in scrambled order - what the GPU work queue does to the driver."""

    def __init__(self, scorer, seed=0):
        self.sc, self.rng, self.inflight = scorer, np.random.default_rng(seed), []
        self.pumps = 0

    def request(self, col, kernel, callback):
        spec = ModelSpec(kernel, self.sc.X.shape[1], K.GaussianLikelihood(), K.ConstantMean())
        if (col, ks.structure_signature(spec)) in self.sc.cache:
            callback(self.sc([(col, kernel)])[0])
            return
        self.inflight.append([int(self.rng.integers(1, 5)), col, kernel, callback])

    def busy(self):
        return bool(self.inflight)

    def pump(self):
        self.pumps += 1
        self.rng.shuffle(self.inflight)
        due = [t for t in self.inflight if t[0] <= 1]
        self.inflight = [[t[0] - 1] + t[1:] for t in self.inflight if t[0] > 1]
        for _, col, kernel, cb in due:
            cb(self.sc([(col, kernel)])[0])

    def close(self):
        pass


def test_unsynchronised_search_equals_lock_step_search():
    """run_searches_async: columns advance whenever THEIR candidates are scored; same fits, so the same trees."""
    from waveome_amd.model_search import GPSearch

    X, Y = search_data()
    gps = GPSearch(X, Y, categorical_vars=["grp"])
    Xn, Yn = gps.X.to_numpy(), gps.Y.to_numpy()
    kl = [K.SquaredExponential(), K.Lin()]
    scorer = OracleScorer(Xn, Yn)
    ref = ks.run_searches(range(3), 3, kl, scorer, cat_vars=[2], max_depth=3, keep_only_best=False)
    fits_lock_step = scorer.n_fits
    scorer2 = OracleScorer(Xn, Yn)
    session = _DelayedOracleSession(scorer2, seed=5)
    got = ks.run_searches_async(range(3), 3, kl, scorer2, cat_vars=[2], max_depth=3, keep_only_best=False,
                                session=session)
    assert session.pumps > 3 and scorer2.n_fits == fits_lock_step
    for c in range(3):
        assert got[c]["best_model"] == ref[c]["best_model"]
        assert set(got[c]["models"]) == set(ref[c]["models"]) and got[c]["edges"] == ref[c]["edges"]
        for name in ref[c]["models"]:
            assert got[c]["models"][name]["bic"] == ref[c]["models"][name]["bic"]
            assert got[c]["models"][name]["parent"] == ref[c]["models"][name]["parent"]


@pytest.mark.gpu
def test_gpu_search_matches_oracle_scored_search():
    from waveome_amd.model_search import GPSearch

    X, Y = search_data()
    kl = [K.SquaredExponential(), K.Lin()]
    gps = GPSearch(X, Y, categorical_vars=["grp"])
    gps.run_search(kernels=kl, max_depth=3, keep_only_best=False)
    Xn, Yn = gps.X.to_numpy(), gps.Y.to_numpy()
    ref = ks.run_searches(range(3), 3, kl, OracleScorer(Xn, Yn), cat_vars=[2], max_depth=3, keep_only_best=False)
    for p, feat in enumerate(gps.out_names):
        got = gps.search_info[feat]
        assert got["best_model"] == ref[p]["best_model"]
        assert set(got["models"]) == set(ref[p]["models"])
        bics = np.array([[got["models"][n]["bic"], ref[p]["models"][n]["bic"]] for n in got["models"]])
        close = np.abs(bics[:, 0] - bics[:, 1]) <= 0.02 + 1e-6 * np.abs(bics[:, 1])
        assert close.mean() >= 0.9, bics[~close]
        assert abs(got["models"][got["best_model"]]["bic"] - ref[p]["models"][ref[p]["best_model"]]["bic"]) <= 0.02
        m = gps.models[feat]
        assert m.kernel_name == got["best_model"]
        # the stored model carries the fitted parameters: its objective is the recorded one
        assert abs(m.training_loss() - got["models"][got["best_model"]]["fit"]["loss"]) < 1e-6
        assert len(got["var_exp"]) == (len(m.kernel.kernels) if m.kernel.name == "sum" else 1) + 1
    assert gps.search_stats["fits"] > 30
    # the default driver is the unsynchronised one; the level-by-level driver must give the same search, bit for bit
    os.environ["WV_SEARCH_ASYNC"] = "0"
    try:
        gps2 = GPSearch(X, Y, categorical_vars=["grp"])
        gps2.run_search(kernels=kl, max_depth=3, keep_only_best=False)
    finally:
        del os.environ["WV_SEARCH_ASYNC"]
    for feat in gps.out_names:
        a, b = gps.search_info[feat], gps2.search_info[feat]
        assert a["best_model"] == b["best_model"] and set(a["models"]) == set(b["models"])
        for n in a["models"]:
            assert a["models"][n]["fit"]["loss"] == b["models"][n]["fit"]["loss"], n


# ---------------------------------------------------------------------------------------------
# f-3: folds, random restarts and the k-fold search over penalisation factors
# ---------------------------------------------------------------------------------------------
def test_make_folds_partitions_rows_and_keeps_units_together():
    from waveome_amd.regularization import make_folds

    X = np.column_stack([np.arange(11.0), np.repeat(np.arange(6.0), 2)[:11]])
    folds = make_folds(X, None, k_fold=3, random_seed=7)
    assert sorted(np.concatenate(folds).tolist()) == list(range(11))
    assert [len(f) for f in folds] == [4, 4, 3]  # divmod split (regularization.py:266-270)
    # the documented recipe: seed, shuffle arange, split
    np.random.seed(7)
    idx = np.arange(11)
    np.random.shuffle(idx)
    assert folds[0].tolist() == idx[:4].tolist() and folds[2].tolist() == idx[8:].tolist()
    ufolds = make_folds(X, 1, k_fold=3, random_seed=7)
    assert sorted(np.concatenate(ufolds).tolist()) == list(range(11))
    for f in ufolds:  # a unit's rows never straddle folds
        units = set(X[f, 1])
        assert all(set(X[g, 1]).isdisjoint(units) for g in ufolds if g is not f)
    with pytest.raises(AssertionError):
        make_folds(X[:4], 1, k_fold=3)


def _search_model(n=60, seed=11):
    from waveome_amd.model_classes import PSVGP

    rng = np.random.default_rng(seed)
    X = np.column_stack([rng.normal(size=n), rng.normal(size=n)])
    y = np.sin(1.5 * X[:, 0]) + 0.2 * rng.normal(size=n)
    kern = K.Sum([K.SquaredExponential(active_dims=[0]), K.SquaredExponential(active_dims=[1])])
    return PSVGP(X, y[:, None], kernel=kern, penalized_options={"penalization_factor": 1.0}), X, y


@pytest.mark.gpu
def test_random_restart_optimize_keeps_the_best_lml():
    m, X, y = _search_model()
    m.random_restart_optimize(num_restart=3, randomize_kwargs={"random_seed": 5})
    starts = m._restart_points(3, random_seed=5)
    spec = m._spec().to_dict()
    best = max((O.fit(spec, X, y, u0=u) for u in starts), key=lambda r: r["lml"])
    assert abs(m.fit_result["lml"] - best["lml"]) <= 1e-6 * abs(best["lml"])
    assert abs(m.maximum_log_likelihood_objective() - best["lml"]) <= 1e-6 * abs(best["lml"])


@pytest.mark.gpu
def test_penalization_search_matches_oracle_table():
    from waveome_amd.regularization import make_folds

    m, X, y = _search_model()
    factors = [0.0, 1.0, 30.0]
    m.penalization_search(penalization_factor_list=factors, k_fold=3, num_restart=2, random_seed=3,
                          fit_best=True, selection_type="se")
    tab = m.penalization_search_results
    assert tab.shape == (9, 3)
    folds = make_folds(X, None, 3, 3)
    starts = m._restart_points(2, random_seed=3)
    want = []
    for h, fold in enumerate(folds):
        Xtr, ytr = np.delete(X, fold, axis=0), np.delete(y, fold)
        for pf in factors:
            spec = m._twin(Xtr, ytr[:, None], pf)._spec().to_dict()
            r = max((O.fit(spec, Xtr, ytr, u0=u) for u in starts), key=lambda r_: r_["lml"])
            mu, var = O.predict(spec, r["theta"], Xtr, ytr, X[fold], include_noise=True)
            ld = -0.5 * (np.log(2 * np.pi) + np.log(var) + (y[fold] - mu) ** 2 / var)
            want.append([pf, h, ld.mean()])
    want = np.array(want)
    got = tab[np.lexsort((tab[:, 0], tab[:, 1]))]
    np.testing.assert_array_equal(got[:, :2], want[:, :2])
    np.testing.assert_allclose(got[:, 2], want[:, 2], rtol=2e-4, atol=2e-4)
    # selection rule: best mean minus one standard error
    score = {f: want[want[:, 0] == f, 2].mean() - want[want[:, 0] == f, 2].std() / np.sqrt(3) for f in factors}
    assert m.penalization_factor == max(score, key=score.get)
    assert m.fit_result is not None and np.isfinite(m.fit_result["lml"])


@pytest.mark.gpu
def test_run_search_with_random_restarts_never_scores_worse_lml():
    """num_restart > 1: every candidate is fitted from seeded random starts and the best LML is kept
    (model_search.py:2283-2288); the chosen fits are reproducible for a seed."""
    from waveome_amd.model_search import GPSearch

    X, Y = search_data(n=40)
    kl = [K.SquaredExponential(), K.Lin()]
    runs = []
    for _ in range(2):
        gps = GPSearch(X, Y, categorical_vars=["grp"])
        gps.run_search(kernels=kl, max_depth=1, num_restart=3, random_seed=4, keep_only_best=False)
        runs.append({f: {n: m["fit"]["lml"] for n, m in gps.search_info[f]["models"].items()} for f in gps.out_names})
    assert runs[0] == runs[1]
    gps1 = GPSearch(X, Y, categorical_vars=["grp"])
    gps1.run_search(kernels=kl, max_depth=1, keep_only_best=False)
    better = total = 0
    for f in gps1.out_names:
        for n, m in gps1.search_info[f]["models"].items():
            total += 1
            better += runs[0][f][n] >= m["fit"]["lml"] - 1e-6 * abs(m["fit"]["lml"]) - 1e-3
    assert better >= 0.8 * total  # random starts mostly find the default start's optimum or a better one


def _name_of(kern):
    if kern.name == "sum":
        return "+".join(_name_of(k) for k in kern.kernels)
    if kern.name == "product":
        return "*".join(_name_of(k) for k in kern.kernels)
    return ks.kernel_info(kern)


@pytest.mark.filterwarnings("ignore:invalid value encountered")  # the NumPy oracle at a collapsing lengthscale
def test_deep_search_keeps_names_structures_and_lineage_consistent():
    """Default kernel list, depth 3, scored by the CPU oracle on a tiny data set: every recorded model's name
    spells its kernel exactly, no structure is fitted twice under two spellings, every parent exists, products
    stay two-way, and the reported best model has the lowest (bic, depth, name) triple."""
    from waveome_amd.model_search import GPSearch

    X, Y = search_data(n=24, seed=5)
    gps = GPSearch(X, Y[["smooth_t"]], categorical_vars=["grp"])
    Xn, Yn = gps.X.to_numpy(), gps.Y.to_numpy()
    scorer = OracleScorer(Xn, Yn)
    res = ks.run_searches([0], 3, default_kernels(), scorer, cat_vars=[2], max_depth=3, keep_only_best=False)[0]
    models = res["models"]
    assert len(models) > 40 and scorer.n_fits >= len(models)
    canon = set()
    for name, m in models.items():
        assert _name_of(m["kernel"]) == name
        assert all(part.count("*") <= 1 for part in name.split("+"))
        key = frozenset("".join(sorted(part)) for part in name.split("+"))
        assert key not in canon, f"{name} duplicates an earlier structure"
        canon.add(key)
        assert m["parent"] == "None" or m["parent"] in models
        assert 1 <= m["depth"] <= 3 and np.isfinite(m["bic"])
        if "*" in name:  # the factor that joined a product has its variance frozen at 1 (model_search.py:2469-2472)
            for comp in (m["kernel"].kernels if m["kernel"].name == "sum" else [m["kernel"]]):
                if comp.name == "product":
                    frozen = [f for f in comp.kernels
                              if not (f.base_kernel.variance if f.name == "periodic" else f.variance).trainable]
                    assert len(frozen) >= 1
    assert res["best_model"] == min((m["bic"], m["depth"], n) for n, m in models.items())[2]
    for parent, child in res["edges"]:
        assert parent in models


# ---------------------------------------------------------------------------------------------
# the unsynchronised search sharded over ranks by COLUMN (kernel_search.run_searches_sharded)
# ---------------------------------------------------------------------------------------------
_SHARDED_WORKER = r'''
import os, pickle, sys
sys.path.insert(0, os.environ["WV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["WV_ROOT"], "tests"))
import numpy as np
import torch.distributed as dist
import test_search as T
from waveome_amd import kernel_search as ks, kernels as K, parallel
from waveome_amd.model_search import GPSearch
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = parallel.dist_info()
X, Y = T.search_data()
Y["lin_dose_b"] = 0.7 * Y["lin_dose"] + 0.3 * Y["noise"]; Y["smooth_b"] = Y["smooth_t"] * 1.3
gps = GPSearch(X, Y, categorical_vars=["grp"])
Xn, Yn = gps.X.to_numpy(), gps.Y.to_numpy()
scorer = T.OracleScorer(Xn, Yn)
session = T._DelayedOracleSession(scorer, seed=7 + rank)       # each rank's queue answers in its own scrambled order
got = ks.run_searches_sharded(Yn.shape[1], 3, [K.SquaredExponential(), K.Lin()], scorer, cat_vars=[2], session=session,
                              max_depth=3, keep_only_best=False)
stats = parallel.last_map_stats()
slim = {c: {"best": r["best_model"], "edges": r["edges"],
            "models": {n: (m["bic"], m["parent"], m["depth"]) for n, m in r["models"].items()}} for c, r in got.items()}
with open(os.environ["WV_OUT"] + f".{rank}", "wb") as fh:
    pickle.dump({"res": slim, "stats": stats, "own_fits": scorer.n_fits, "totals": scorer.job_totals}, fh)
dist.barrier()
dist.destroy_process_group()
'''


def test_column_sharded_unsynchronised_search_equals_single_process(tmp_path):
    """gloo, world size 2, the CPU oracle as scorer behind a delaying / scrambling session: every column's search runs
    on one rank, columns are dealt on demand, the gathered result on BOTH ranks equals the single-process search."""
    import pickle
    import socket
    import subprocess

    from waveome_amd.model_search import GPSearch

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "sharded")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), WV_ROOT=str(ROOT), WV_OUT=out, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", _SHARDED_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, log.decode()[-3000:]
    X, Y = search_data()
    Y["lin_dose_b"] = 0.7 * Y["lin_dose"] + 0.3 * Y["noise"]
    Y["smooth_b"] = Y["smooth_t"] * 1.3
    gps = GPSearch(X, Y, categorical_vars=["grp"])
    Xn, Yn = gps.X.to_numpy(), gps.Y.to_numpy()
    ref_scorer = OracleScorer(Xn, Yn)
    ref = ks.run_searches(range(5), 3, [K.SquaredExponential(), K.Lin()], ref_scorer, cat_vars=[2], max_depth=3,
                          keep_only_best=False)
    parts = [pickle.load(open(out + f".{r}", "rb")) for r in range(2)]
    assert parts[0]["res"] == parts[1]["res"]                       # every rank holds the whole job's result
    for c in range(5):
        got = parts[0]["res"][c]
        assert got["best"] == ref[c]["best_model"] and got["edges"] == ref[c]["edges"]
        assert got["models"] == {n: (m["bic"], m["parent"], m["depth"]) for n, m in ref[c]["models"].items()}
    # the columns were split (nobody searched everything), the fit counts add up, the imbalance figure is there
    assert all(0 < p["own_fits"] < ref_scorer.n_fits for p in parts)
    assert parts[0]["own_fits"] + parts[1]["own_fits"] == ref_scorer.n_fits == parts[0]["totals"]["fits"]
    st = parts[0]["stats"]
    assert st["mode"] == "dynamic-columns" and sorted(st["tasks_per_rank"]) == [2, 3]
    assert st["rank_time_max_over_mean"] >= 1.0
