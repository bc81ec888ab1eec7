"""Pins the CPU oracle: reference known answers (KAT-1/2/3 from the reference's notebooks),
finite-difference gradients for every kernel type, algebraic identities, Horseshoe checks."""
import json
import math
import os

import numpy as np
import pytest
import scipy.integrate

from oracle import gp_oracle as o


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "kat.json")) as fh:
        return json.load(fh)


def _kat_problem(c):
    X = np.array(c["X"]).reshape(-1, 1)
    Y = np.array(c["Y"])
    spec = o.make_spec(1, [[(c["kernel"], [0])]], mean=False)
    return spec, X, Y


def test_kat1_value_at_printed_params(kat):
    c = kat["kat1"]
    spec, X, Y = _kat_problem(c)
    e = c["expected"]
    th = np.array([e["variance"], e["lengthscales"], e["noise"]])
    # params are printed to 6 s.f.; LML is flat at the optimum so it agrees to ~1e-11 rel
    assert o.lml(spec, th, X, Y) == pytest.approx(e["lml"], rel=5e-11)


def test_kat1_fit_reproduces_notebook(kat):
    c = kat["kat1"]
    spec, X, Y = _kat_problem(c)
    r = o.fit(spec, X, Y, maxiter=100)
    e = c["expected"]
    assert r["lml"] == pytest.approx(e["lml"], rel=1e-12)
    assert r["theta"][0] == pytest.approx(e["variance"], rel=5e-6)
    assert r["theta"][1] == pytest.approx(e["lengthscales"], rel=5e-6)
    assert r["theta"][2] == pytest.approx(e["noise"], rel=5e-6)


def test_kat1_full_precision_parameters(kat):
    """gpflow_basic_colab.ipynb cell 13 prints the fitted parameters at full precision: SciPy L-BFGS-B on the oracle
    walks GPflow's 100-iteration path to the same point - the two objectives are the same function to rounding."""
    c = kat["kat1"]
    spec, X, Y = _kat_problem(c)
    full = c["expected_full"]
    th = np.array([full["variance"], full["lengthscales"], full["noise"]])
    assert o.lml(spec, th, X, Y) == pytest.approx(c["expected"]["lml"], abs=2e-12)
    r = o.fit(spec, X, Y, maxiter=100)
    np.testing.assert_allclose(r["theta"], th, rtol=1e-9)


def test_kat2_matern52_fit(kat):
    c = kat["kat2"]
    spec, X, Y = _kat_problem(c)
    r = o.fit(spec, X, Y, maxiter=100)
    e = c["expected"]
    assert r["theta"][0] == pytest.approx(e["variance"], rel=5e-6)
    assert r["theta"][1] == pytest.approx(e["lengthscales"], rel=5e-6)
    assert r["theta"][2] == pytest.approx(e["noise"], rel=5e-6)


def test_kat3_matern12_two_decimals(kat):
    c = kat["kat3"]
    spec, X, Y = _kat_problem(c)
    r = o.fit(spec, X, Y, maxiter=100)
    assert abs(r["lml"] - c["expected"]["lml_2dp"]) < 0.01


def test_kat5_periodic_three_decimals(kat):
    """The reference's stored Periodic answer ('# Per = -12.242', gpflow_basic_colab.ipynb cell 7): the path from the
    GPflow defaults is deterministic, the oracle lands on the notebook's number to its printed digits."""
    c = kat["kat5"]
    spec, X, Y = _kat_problem(c)
    r = o.fit(spec, X, Y, maxiter=100)
    assert abs(r["lml"] - c["expected"]["lml_3dp"]) < 1e-3
    rec = c["oracle_record"]
    assert r["lml"] == pytest.approx(rec["lml"], rel=1e-9)
    assert (r["nit"], r["nfev"]) == (rec["nit"], rec["nfev"])
    np.testing.assert_allclose(r["theta"][:3], rec["theta"][:3], rtol=1e-5)


def _kat8(c):
    X = np.column_stack([c["unit"], c["treatment"], c["time"]]).astype(float)
    y = np.asarray(c["y1_obs"], float)
    spec = o.make_spec(3, [[("m12", [2])], [("cat", [0])]], mean=False)
    t = c["theta"]
    theta = np.array([t["matern12.variance"], t["matern12.lengthscales"], t["categorical.variance"], t["noise"]])
    return spec, theta, X, y


def test_kat8_value_at_fixed_parameters_matern12_plus_categorical(kat):
    """data_generation.ipynb cell 14: the reference evaluates, with SciPy's multivariate normal, the log-likelihood of
    the y1_obs column of its own simulated_data.csv under Matern12(1, 1)[time] + Categorical(2)[unit] + 0.01 I and stores
    k1_bic = 3 ln 126 - 2 loglik = 169.71614261744028.  A value at fixed parameters - no optimiser between the stored
    number and the Gram matrix: Matern-1/2, a Categorical term's VALUE, their Sum and the LML formula to 1e-13."""
    c = kat["kat8"]
    spec, theta, X, y = _kat8(c)
    e = c["expected"]
    assert len(y) == e["n"] and sum(p["trainable"] for p in spec["params"]) - 1 == e["k"]   # k1's own parameters
    ll = o.lml(spec, theta, X, y)
    assert e["k"] * np.log(e["n"]) - 2 * ll == pytest.approx(e["k1_bic"], rel=1e-13)
    # the same number from the definition the notebook used
    from scipy.stats import multivariate_normal
    K = o.gram(spec, theta, X) + theta[3] * np.eye(len(y))
    assert multivariate_normal(mean=np.zeros(len(y)), cov=K).logpdf(y) == pytest.approx(ll, rel=1e-12)
    # it discriminates: the Categorical variance alone moves it by whole nats
    th2 = theta.copy(); th2[2] = 1.0
    assert abs(o.lml(spec, th2, X, y) - ll) > 1.0


def test_kat6_constant_mean_se_bounds_from_the_stored_svgp_fits(kat):
    """simple_regression_different_models.ipynb cells 1-5, 7 (BaseGP = SVGP with Z = X, Constant mean, SE; N = 100):
    the notebook stores two parameter tables at full precision with the ELBO each had reached, and the BIC of the
    second.  The SVGP's q had not fully converged (its ELBO sits 0.008 under the collapsed bound), so the numbers
    bound the exact LML instead of pinning it:  stored ELBO <= LML(stored theta) <= max LML, gaps of 1e-4 relative."""
    c = kat["kat6"]
    X = np.asarray(c["X"], float)[:, None]
    Y = np.asarray(c["Y"], float)
    spec = o.make_spec(1, [[("se", [0])]], mean=True)
    names = ("kernel.variance", "kernel.lengthscales", "likelihood.variance", "mean_function.c")
    exp, rec = c["expected"], c["oracle_record"]
    th1 = np.array([exp["first_fit"][k] for k in names])
    th2 = np.array([exp["after_restarts"][k] for k in names])
    l1, l2 = o.lml(spec, th1, X, Y), o.lml(spec, th2, X, Y)
    assert 0.0 <= l1 - (-exp["first_fit"]["training_loss_round_1500"]) <= 0.03          # 72.4743 vs 72.4485
    elbo2 = (2 * exp["n_parameter_objects"] - exp["after_restarts"]["bic"]) / 2                # 73.22067...
    assert elbo2 >= -exp["after_restarts"]["training_loss_round_1000"]                        # later round, higher ELBO
    assert 0.0 <= l2 - elbo2 <= 0.012                                                          # 73.2301 vs 73.2207
    best = o.fit(spec, X, Y)
    assert 0.0 <= best["lml"] - l2 <= 0.06                                                     # 73.2726
    # the BIC rule itself (2 k - 2 log posterior, k = 6 Parameter objects): our LML in it lands within 0.02 of the print
    assert abs((2 * 6 - 2 * l2) - exp["after_restarts"]["bic"]) <= 0.02
    assert [round(th2[0], 3), round(th2[2], 3)] == exp["after_restarts"]["variance_explained"]
    assert l1 == pytest.approx(rec["lml_at_first_fit"], rel=1e-10)
    assert l2 == pytest.approx(rec["lml_after_restarts"], rel=1e-10)
    assert best["lml"] == pytest.approx(rec["lml_optimum"], rel=1e-8)


def test_transforms_roundtrip():
    for x in [1e-8, 1e-3, 0.5, 1.0, 7.0, 50.0, 800.0]:
        assert o.softplus(o.inv_softplus(x)) == pytest.approx(x, rel=1e-12)
    assert o.constrain_one("softplus_shift", o.unconstrain_one("softplus_shift", 1.0)) == pytest.approx(1.0)
    assert o.constrain_one("softplus_shift", -800.0) == pytest.approx(1e-6)


def _rand_problem(seed, N=40, D=4):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, D))
    X[:, 3] = rng.integers(0, 3, N).astype(float)
    y = np.sin(X[:, 0]) + 0.3 * X[:, 1] + 0.5 * (X[:, 3] == 1) + 0.1 * rng.normal(size=N)
    return X, y


ALL_TERMS = [
    [[("se", [0])]],
    [[("se", [0, 1, 2])]],
    [[("m12", [1])]],
    [[("m32", [1])]],
    [[("m52", [0, 1])]],
    [[("periodic", [0])]],
    [[("lin", [2])]],
    [[("poly", [2], {"degree": 3})]],
    [[("const", [0])]],
    [[("cat", [3])]],
    [[("cat", [3])], [("se", [0])], [("se", [1])], [("cat", [3], {"freeze_variance": True}), ("se", [0])],
     [("cat", [3], {"freeze_variance": True}), ("periodic", [1])]],
    [[("se", [0]), ("m32", [1])], [("lin", [2]), ("periodic", [0])], [("empty", [0])]],
]


@pytest.mark.parametrize("terms", ALL_TERMS)
def test_gradient_matches_finite_differences(terms):
    X, y = _rand_problem(3)
    spec = o.make_spec(4, terms, mean=True, prior_scale=1.0)
    rng = np.random.default_rng(5)
    u = o.pack_u(spec, o.theta_init(spec)) + 0.3 * rng.normal(size=len(o.trainable_index(spec)))
    f0, g0 = o.objective(spec, u, X, y)
    for k in range(len(u)):
        h = 1e-6
        up, um = u.copy(), u.copy()
        up[k] += h
        um[k] -= h
        fd = (o.objective(spec, up, X, y)[0] - o.objective(spec, um, X, y)[0]) / (2 * h)
        assert g0[k] == pytest.approx(fd, rel=2e-6, abs=2e-7), (k, spec["params"][o.trainable_index(spec)[k]]["name"])


def test_gram_symmetric_psd_and_sum_product_algebra():
    X, _ = _rand_problem(7)
    a = o.make_spec(4, [[("se", [0])]])
    b = o.make_spec(4, [[("m32", [1])]])
    ab = o.make_spec(4, [[("se", [0])], [("m32", [1])]])
    axb = o.make_spec(4, [[("se", [0]), ("m32", [1])]])
    Ka = o.gram(a, o.theta_init(a), X)
    Kb = o.gram(b, o.theta_init(b), X)
    np.testing.assert_allclose(o.gram(ab, o.theta_init(ab), X), Ka + Kb, rtol=1e-15)
    np.testing.assert_allclose(o.gram(axb, o.theta_init(axb), X), Ka * Kb, rtol=1e-15)
    np.testing.assert_array_equal(Ka, Ka.T)
    assert np.linalg.eigvalsh(Ka + 1e-6 * np.eye(len(X))).min() > 0


def test_categorical_equals_lin_on_one_hot():
    # waveome/kernels.py:95-117 Categorical == sum of Lin kernels over one-hot columns
    X, _ = _rand_problem(9)
    cat = o.make_spec(4, [[("cat", [3])]])
    Kc = o.gram(cat, o.theta_init(cat), X)
    onehot = np.stack([(np.rint(X[:, 3]) == c).astype(float) for c in range(3)], axis=1)
    np.testing.assert_array_equal(Kc, onehot @ onehot.T)


def test_categorical_rounds_half_to_even():
    X = np.array([[0.5], [1.5], [2.5], [0.0], [2.0]])
    cat = o.make_spec(1, [[("cat", [0])]])
    K = o.gram(cat, o.theta_init(cat), X)
    # rint: 0.5->0, 1.5->2, 2.5->2
    assert K[0, 3] == 1 and K[1, 4] == 1 and K[2, 4] == 1 and K[0, 1] == 0


def test_periodic_tends_to_se_for_long_period():
    rng = np.random.default_rng(1)
    X = rng.uniform(-1, 1, size=(30, 1))
    per = o.make_spec(1, [[("periodic", [0])]])
    se = o.make_spec(1, [[("se", [0])]])
    P = 1e4
    thp = o.theta_init(per)
    thp[1] = 1.0 * np.pi / P  # sin(pi d/p)/ell ~ d * (pi/p) / ell -> ell_se = ell*p/pi
    thp[2] = P
    Kp = o.gram(per, thp, X)
    Ks = o.gram(se, o.theta_init(se), X)
    np.testing.assert_allclose(Kp, Ks, rtol=1e-6)  # limit error ~ (pi d/p)^2/6


def test_lml_matches_scipy_multivariate_normal():
    from scipy.stats import multivariate_normal

    X, y = _rand_problem(11, N=25)
    spec = o.make_spec(4, [[("se", [0])], [("cat", [3])]])
    th = o.theta_init(spec)
    th[spec["mean"]] = 0.3
    Ky = o.gram(spec, th, X) + th[spec["noise"]] * np.eye(25)
    ref = multivariate_normal(mean=np.full(25, 0.3), cov=Ky).logpdf(y)
    assert o.lml(spec, th, X, y) == pytest.approx(ref, rel=1e-12)


def test_low_rank_kernels_against_their_closed_form():
    """Lin, Poly, Constant and Categorical Gram matrices have low rank, so the LML of their sums has a closed form
    through a small capacitance matrix (tests/big_n_closed_form.py): an exact, oracle-independent answer for the two
    kernels no stored output of the reference pins (Lin: waveome/kernels.py:19-32, Poly: :56-73)."""
    import big_n_closed_form as B

    rng = np.random.default_rng(0)
    N = 300
    x = rng.uniform(-1.5, 1.5, N)
    y = 0.4 * x + 0.3 * x * x - 0.2 + 0.2 * rng.normal(size=N)
    spec = o.make_spec(1, [[("lin", [0])], [("poly", [0], {"degree": 2})], [("const", [0])]], mean=False)
    names = [p["name"] for p in spec["params"]]
    assert [n.split(".")[-1] for n in names] == ["variance", "variance", "offset", "variance", "variance"]
    theta = np.array([0.7, 0.5, 0.8, 0.3, 0.05])
    assert o.lml(spec, theta, x[:, None], y) == pytest.approx(B.closed_form_lin_poly(0.7, 0.5, 0.8, 0.3, 0.05, x, y), rel=1e-11)
    lab = rng.integers(0, 2, N).astype(float)
    X = np.column_stack([x, lab])
    spec2 = o.make_spec(2, [[("const", [0, 1])], [("cat", [1])]], mean=False)
    assert o.lml(spec2, np.array([0.8, 0.4, 0.09]), X, y) == pytest.approx(B.closed_form(0.8, 0.4, 0.09, lab, y), rel=1e-11)


@pytest.mark.parametrize("kind", ["m12", "m32", "m52"])
def test_matern_family_against_the_state_space_likelihood(kind):
    """The Matern kernels against a derivation that shares nothing with the oracle: the Kalman-filter likelihood of
    the equivalent stochastic differential equation (tests/statespace_matern.py).  Matern-3/2 has no stored value
    anywhere in the reference (parity unpinned); this is its independent check (Matern-1/2 and -5/2 also have KAT-3 / KAT-2)."""
    import statespace_matern as S

    rng = np.random.default_rng(7)
    N = 400
    x = np.sort(rng.uniform(0, 12, N))
    y = np.sin(x) + 0.3 * np.cos(2.3 * x) + 0.2 * rng.normal(size=N)
    spec = o.make_spec(1, [[(kind, [0])]], mean=True)
    for var, ell, s2, c in [(1.0, 1.0, 0.1, 0.0), (2.3, 0.35, 0.02, 0.4), (0.6, 4.0, 0.5, -0.2)]:
        want = S.kalman_lml(kind, var, ell, s2, x, y, mean=c)
        assert o.lml(spec, np.array([var, ell, s2, c]), x[:, None], y) == pytest.approx(want, rel=1e-9)


def test_periodic_kernel_against_its_bessel_cosine_series():
    """The Periodic(SE) kernel - half of the headline structure, pinned by the reference to 3 dp (KAT-5) - against an
    exact, independent derivation: its cosine series with modified-Bessel weights is a low-rank kernel whose LML has a
    closed form (oracle/closed_forms.py::periodic_features)."""
    import big_n_closed_form as B

    rng = np.random.default_rng(4)
    N = 300
    x = rng.uniform(-3, 3, N)
    y = np.sin(2.1 * x) + 0.3 * x + 0.2 * rng.normal(size=N)
    spec = o.make_spec(1, [[("periodic", [0])], [("lin", [0])]], mean=True)
    assert [p["name"].split(".")[-1] for p in spec["params"]] == ["variance", "lengthscales", "period", "variance", "variance", "c"]
    for v, ell, per, vl, s2, c in [(1.0, 1.0, 1.0, 1.0, 1.0, 0.0), (1.8, 0.45, 2.7, 0.2, 0.03, 0.3), (0.5, 0.12, 1.3, 0.05, 0.1, -0.1)]:
        want = B.closed_form_periodic(v, ell, per, vl, s2, x, y, mean=c)
        assert o.lml(spec, np.array([v, ell, per, vl, s2, c]), x[:, None], y) == pytest.approx(want, rel=1e-10)
    # the Gram matrix itself from the series
    U, d = B.periodic_features(x, 1.8, 0.45, 2.7)
    Kser = (U * d) @ U.T
    Kor = o.gram(o.make_spec(1, [[("periodic", [0])]], mean=False), np.array([1.8, 0.45, 2.7, 1.0]), x[:, None])
    np.testing.assert_allclose(Kser, Kor, rtol=0, atol=1e-13)


def test_headline_structure_against_its_low_rank_closed_form():
    """The BASELINE headline structure sum_d (SE[d] + Periodic(SE)[d]) + Constant mean is, to rounding, a kernel of rank
    a few hundred on bounded inputs (SE: Taylor feature map; Periodic: Bessel cosine series), so its exact LML follows
    from one capacitance matrix at ANY N - an answer for the metric's own objective that never forms a Gram matrix
    (oracle/closed_forms.py::headline_closed_form).  The SE Gram matrix itself against its series, too."""
    import bench
    import big_n_closed_form as B

    saved = bench.N_HEAD
    bench.N_HEAD = 400
    try:
        X, Y = bench.make_inputs(2)
    finally:
        bench.N_HEAD = saved
    ms = bench.headline_spec()
    d = ms.to_dict()
    rng = np.random.default_rng(0)
    for col, theta in ((0, ms.theta()), (1, ms.constrain(ms.unconstrain() + 0.4 * rng.normal(size=ms.n_trainable)))):
        assert o.lml(d, theta, X, Y[col]) == pytest.approx(B.headline_closed_form(theta, X, Y[col]), rel=1e-12)
    x = X[:, 0]
    for ell in (0.3, 1.0, 2.5):
        U, w = B.se_features(x, 1.3, ell)
        Kor = o.gram(o.make_spec(1, [[("se", [0])]], mean=False), np.array([1.3, ell, 1.0]), x[:, None])
        np.testing.assert_allclose((U * w) @ U.T, Kor, rtol=0, atol=2e-13)


def test_random_low_rank_structures_against_their_closed_form():
    """Seeded random sums of products over the kernels that have a finite feature map - SE (one and two columns),
    Periodic, Lin, Poly (degree 2, 3), Categorical, Constant - : the oracle's LML against the capacitance-matrix closed
    form (oracle/closed_forms.py::spec_closed_form; features of a product = row-wise Kronecker product).  Two
    derivations, no shared code: what makes either wrong makes them disagree.  (Poly and the value of a surviving
    Product term have no stored answer in the reference - parity unpinned; this is their independent check.)"""
    from oracle import closed_forms as CF
    from waveome_amd import ModelSpec, kernels as K

    kinds = [lambda r: K.SquaredExponential(active_dims=[int(r.integers(0, 3))], variance=r.uniform(.4, 2), lengthscales=r.uniform(.5, 2)),
             lambda r: K.Periodic(K.SquaredExponential(active_dims=[int(r.integers(0, 3))], variance=r.uniform(.4, 2),
                                                       lengthscales=r.uniform(.4, 2)), period=r.uniform(.5, 3)),
             lambda r: K.Lin(active_dims=[int(r.integers(0, 3))], variance=r.uniform(.2, 1)),
             lambda r: K.Poly(active_dims=[int(r.integers(0, 3))], variance=r.uniform(.2, 1), offset=r.uniform(.3, 1.5),
                              degree=int(r.integers(2, 4))),
             lambda r: K.Categorical(active_dims=[3], variance=r.uniform(.2, 1)),
             lambda r: K.Constant(variance=r.uniform(.2, 1)),
             lambda r: K.SquaredExponential(active_dims=[0, 1], variance=r.uniform(.4, 2), lengthscales=r.uniform(.7, 2))]
    done = 0
    for seed in range(30):
        r = np.random.default_rng(seed)
        terms = []
        for _ in range(int(r.integers(1, 5))):
            fs = [kinds[int(r.integers(len(kinds)))](r) for _ in range(int(r.integers(1, 3)))]
            terms.append(fs[0] if len(fs) == 1 else K.Product(fs))
        ms = ModelSpec(terms[0] if len(terms) == 1 else K.Sum(terms), 4, K.GaussianLikelihood(variance=r.uniform(.02, .5)),
                       K.ConstantMean(0.3 * r.normal()))
        rr = np.random.default_rng(1000 + seed)
        X = rr.normal(size=(150, 4))
        X[:, 3] = rr.integers(0, 3, 150)
        y = np.sin(X[:, 0]) + 0.3 * X[:, 1] + 0.2 * rr.normal(size=150)
        d, th = ms.to_dict(), ms.theta()
        try:
            want = CF.spec_closed_form(d, th, X, y, max_rank=2500)
        except ValueError:      # products of wide feature maps: rank beyond what a closed form is practical for
            continue
        assert o.lml(d, th, X, y) == pytest.approx(want, rel=1e-10), seed
        done += 1
    assert done >= 15


def test_closed_form_gradient_of_the_variance_parameters():
    """oracle/closed_forms.py::spec_closed_form_grad: exact derivatives of the Gram-free LML with respect to every
    variance parameter, the noise and the mean - against the oracle's dense gradient on a full_kernel_build structure
    (SE, Periodic, Lin base kernels, categorical x numeric products) and on Poly terms."""
    from oracle import closed_forms as CF
    from waveome_amd import ModelSpec, kernels as K
    from waveome_amd.regularization import full_kernel_build

    rng = np.random.default_rng(1)
    N = 250
    X = rng.normal(size=(N, 5))
    X[:, 4] = rng.integers(0, 5, N)
    y = np.sin(X[:, 0]) + 0.4 * (X[:, 4] == 2) + 0.2 * rng.normal(size=N)
    kern = full_kernel_build(cat_vars=[4], num_vars=[0, 1, 2, 3], return_sum=True,
                             kerns=[K.SquaredExponential(), K.Periodic(K.SquaredExponential()), K.Lin()])
    poly = K.Sum([K.Poly(active_dims=[0], degree=3, variance=0.6, offset=0.7),
                  K.Product([K.Categorical(active_dims=[4], variance=0.5), K.Poly(active_dims=[1], degree=2)])])
    for kern_, n_exact in ((kern, 39), (poly, 5)):
        ms = ModelSpec(kern_, 5, K.GaussianLikelihood(variance=0.3), K.ConstantMean(0.1))
        d = ms.to_dict()
        th = ms.constrain(ms.unconstrain() + 0.3 * rng.normal(size=ms.n_trainable))
        lml, g = CF.spec_closed_form_grad(d, th, X, y)
        lo, go = o.lml_and_grad(d, th, X, y)
        ok = ~np.isnan(g)
        assert ok.sum() == n_exact and lml == pytest.approx(lo, rel=1e-11)
        np.testing.assert_allclose(g[ok], go[ok], rtol=1e-8, atol=1e-9 * np.max(np.abs(go)))


def test_product_with_a_categorical_factor_is_one_independent_gp_per_level():
    """Categorical[g] * Matern32[x] (the unit x numeric / categorical x numeric terms of full_kernel_build,
    waveome/regularization.py:103-149) is block diagonal in the labels: its LML is the sum over the levels of the LML of
    an independent GP with variance w * v on that level's rows - each computed here by the Kalman filter of
    tests/statespace_matern.py.  An exact, oracle-independent value of a PRODUCT term."""
    import statespace_matern as S

    rng = np.random.default_rng(3)
    N = 360
    x = rng.uniform(0, 10, N)
    g = rng.integers(0, 4, N).astype(float)
    y = np.sin(x + g) + 0.2 * rng.normal(size=N)
    spec = o.make_spec(2, [[("cat", [1]), ("m32", [0])]], mean=True)
    w, v, ell, s2, c = 0.8, 1.6, 0.7, 0.05, 0.1
    names = [p["name"].split(".")[-1] for p in spec["params"]]
    assert names == ["variance", "variance", "lengthscales", "variance", "c"]
    want = sum(S.kalman_lml("m32", w * v, ell, s2, x[g == lev], y[g == lev], mean=c) for lev in range(4))
    assert o.lml(spec, np.array([w, v, ell, s2, c]), np.column_stack([x, g]), y) == pytest.approx(want, rel=1e-9)


def test_horseshoe_bounds_and_integral():
    # Carvalho-Polson-Scott bounds: (K/2) log(1+4/x^2) < p(x) < K log(1+2/x^2), K = (2 pi^3)^-1/2
    Kc = 1.0 / math.sqrt(2.0 * math.pi**3)
    for x in [1e-3, 0.05, 0.3, 1.0, 2.5, 10.0, 80.0]:
        p = math.exp(o.horseshoe_logpdf(x, 1.0))
        assert Kc / 2 * math.log1p(4 / x**2) < p < Kc * math.log1p(2 / x**2)
        # exact marginal: int_0^inf N(x|0,lam^2) 2/(pi(1+lam^2)) dlam
        f = lambda lam: math.exp(-0.5 * (x / lam) ** 2) / (lam * math.sqrt(2 * math.pi)) * 2 / (math.pi * (1 + lam * lam))
        exact = scipy.integrate.quad(f, 0, np.inf, limit=400)[0]
        assert p == pytest.approx(exact, rel=2e-3)
    # scale family: p(x; s) = p(x/s; 1)/s
    assert o.horseshoe_logpdf(0.7, 0.25) == pytest.approx(o.horseshoe_logpdf(2.8, 1.0) - math.log(0.25), rel=1e-13)


def test_horseshoe_derivative():
    for s in [1.0, 0.1, 4.0]:
        for x in [1e-2, 0.3, 1.0, 5.0]:
            h = 1e-6 * x
            fd = (o.horseshoe_logpdf(x + h, s) - o.horseshoe_logpdf(x - h, s)) / (2 * h)
            assert o.horseshoe_dlogpdf(x, s) == pytest.approx(fd, rel=1e-6)


def test_non_pd_reports_inf():
    X = np.zeros((5, 1))
    spec = o.make_spec(1, [[("lin", [0])]], mean=False)
    u = o.pack_u(spec, o.theta_init(spec))
    u[-1] = -1e6  # noise -> floor 1e-6, K = 0 -> still PD (1e-6 I); make it fail with NaN input
    f, _ = o.objective(spec, u, X, np.ones(5))
    assert np.isfinite(f)
