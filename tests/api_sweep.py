"""By-hand robustness sweep of the GPSearch / PSVGP surface on two small real-shaped data sets (the KAT-7 and KAT-4 data):
every public option of the hot path in combinations the unit tests do not reach, followed by every post-fit method on
every model it produced.  Prints one line per step; any exception is a finding.
    python tests/api_sweep.py"""
import json
import os
import sys
import time
import traceback

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from waveome_amd import kernels as K  # noqa: E402
from waveome_amd.model_search import GPSearch  # noqa: E402

kat = json.load(open(os.path.join(HERE, "golden", "kat.json")))
failures = []


def step(name, fn):
    t0 = time.time()
    try:
        out = fn()
        print(f"ok   {name}  ({time.time() - t0:.1f} s){'' if out is None else '  ' + str(out)[:110]}", flush=True)
        return out
    except Exception as e:  # noqa: BLE001
        failures.append(name)
        print(f"FAIL {name}: {type(e).__name__}: {e}", flush=True)
        traceback.print_exc()


def post_fit(tag, gps):
    for out, m in getattr(gps, "models", {}).items():
        if isinstance(m, dict):            # run_search keeps {"models": ..., "best_model": ...} in search_info; models[out] is a PSVGP
            continue
        X, Y = m.data
        step(f"{tag}/{out} predict_f + predict_y", lambda: (m.predict_f()[0].shape, m.predict_y()[1].shape))
        step(f"{tag}/{out} predict_y(new rows)", lambda: m.predict_y(X[:7] + 0.01)[0].shape)
        step(f"{tag}/{out} predict_log_density", lambda: float(m.predict_log_density((X[:9], Y[:9])).sum()))
        step(f"{tag}/{out} calc_metric / densities", lambda: (round(m.calc_metric(), 3), round(m.training_loss(), 3)))
        step(f"{tag}/{out} feature importances", lambda: (m.get_feature_importances(), m.feature_importances)[1])
        step(f"{tag}/{out} q_mu / q_sqrt / Z", lambda: (m.q_mu.shape, m.q_sqrt.shape, m.inducing_variable.Z.numpy().shape))
        step(f"{tag}/{out} kernel name", lambda: m.kernel_name)


def frames(which):
    if which == "kat7":
        c = kat["kat7"]
        X = pd.DataFrame(c["X"], columns=[f"X{i}" for i in range(5)])
        Y = pd.DataFrame({"Y": c["Y"], "Y2": np.asarray(c["Y"]) ** 2})
        return dict(X=X, Y=Y, unit_col="X4", categorical_vars=["X4"], Y_transform="standardize")
    df = pd.DataFrame(kat["kat4"]["columns"])
    return dict(X=df[["person_id", "time", "female"]], Y=df[["outcome1", "outcome2", "outcome3"]], unit_col="person_id",
                categorical_vars=["female"])


kerns4 = [K.SquaredExponential(), K.Matern12(), K.Periodic(K.SquaredExponential()), K.Lin()]
for which in ("kat7", "kat4"):
    kw = frames(which)
    g = step(f"{which} GPSearch()", lambda: GPSearch(**kw))
    if step(f"{which} penalized_optimization()", lambda: g.penalized_optimization(random_seed=1)) is None:
        post_fit(f"{which} default", g)
    g = GPSearch(**kw)
    step(f"{which} penalized_optimization(penalization_factor=None)",
         lambda: g.penalized_optimization(penalization_factor=None, num_factor_iter=3, random_seed=2))
    post_fit(f"{which} iterated factor", g)
    g = GPSearch(**kw)
    step(f"{which} penalized_optimization(num_restart=3, four kernels, unit and categorical interactions)",
         lambda: g.penalized_optimization(num_restart=3, random_seed=3, kernel_options={
             "second_order_numeric": False, "categorical_numeric_interactions": True,
             "unit_numeric_interactions": True, "kerns": kerns4}))
    post_fit(f"{which} restarts", g)
    g = GPSearch(**kw)
    step(f"{which} penalized_optimization(second_order_numeric, SE only)",
         lambda: g.penalized_optimization(random_seed=7, kernel_options={
             "second_order_numeric": True, "categorical_numeric_interactions": True,
             "unit_numeric_interactions": False, "kerns": [K.SquaredExponential()]}))
    post_fit(f"{which} second order", g)
    if which == "kat7":   # 4 numeric columns x 4 kernels with all pairs: more parameters than one device structure holds
        g = GPSearch(**kw)

        def too_big():
            try:
                g.penalized_optimization(kernel_options={"second_order_numeric": True, "categorical_numeric_interactions": True,
                                                         "unit_numeric_interactions": True, "kerns": kerns4})
            except NotImplementedError as e:
                return "refused by name: " + str(e)[:80]
            raise AssertionError("a structure beyond the device limits was accepted")

        step(f"{which} penalized_optimization(everything at once) is refused with the limit named", too_big)
    g = GPSearch(**kw)
    step(f"{which} penalized_optimization(penalization_factor=0)", lambda: g.penalized_optimization(penalization_factor=0.0))
    g = GPSearch(**kw)
    step(f"{which} run_search(depth 3, 2 restarts)", lambda: g.run_search(max_depth=3, num_restart=2, random_seed=4))
    post_fit(f"{which} search", g)
    step(f"{which} search var_exp", lambda: {k: v["var_exp"] for k, v in g.search_info.items()})
    g = GPSearch(**kw)
    cont = g.cont_idx
    structures = {"se": K.Sum([K.SquaredExponential(active_dims=[c]) for c in cont]),
                  "per+cat": K.Sum([K.Periodic(K.SquaredExponential(active_dims=[cont[0]])),
                                    K.Categorical(active_dims=[g.cat_idx[0]])]),
                  "m32*lin": K.Product([K.Matern32(active_dims=[cont[0]]), K.Lin(active_dims=[cont[-1]])])}
    step(f"{which} kernel_sweep(3 structures, 2 restarts)",
         lambda: g.kernel_sweep(structures, num_restart=2, random_seed=5)[["bic", "nit"]].round(2).to_dict())
    m = next(iter(g.sweep_models.values()))
    step(f"{which} PSVGP.random_restart_optimize", lambda: m.random_restart_optimize(num_restart=3))
    step(f"{which} PSVGP.penalization_search",
         lambda: m.penalization_search(penalization_factor_list=[0.0, 1.0, 10.0], k_fold=3, random_seed=6))
    step(f"{which} PSVGP after the search", lambda: (m.penalization_factor, round(m.calc_metric(), 2)))
# ---- awkward inputs: they must fit, or be refused cleanly - never crash, hang or return a finite score for garbage ----
rng = np.random.default_rng(0)


def awkward(name, X, Y, **kw):
    def run():
        g = GPSearch(X=X, Y=Y, **kw)
        g.penalized_optimization(random_seed=1)
        out = {}
        for k, m in g.models.items():
            m.get_feature_importances()
            out[k] = (m.kernel_name, round(float(m.likelihood.variance.value), 6), bool(np.all(np.isfinite(m.predict_y()[0]))))
        return out
    step("awkward: " + name, run)


x = rng.normal(size=40)
awkward("every row twice (K singular up to the noise)",
        pd.DataFrame({"a": np.repeat(x[:20], 2), "b": np.repeat(rng.normal(size=20), 2)}),
        pd.DataFrame({"y": np.repeat(np.sin(x[:20]), 2) + 0.01 * rng.normal(size=40)}))
awkward("three rows", pd.DataFrame({"a": [0.0, 1.0, 2.0]}), pd.DataFrame({"y": [0.1, 0.9, 2.2]}))
awkward("a categorical column with one level",
        pd.DataFrame({"a": x, "g": np.zeros(40)}), pd.DataFrame({"y": np.cos(x)}), categorical_vars=["g"])
awkward("every row its own unit",
        pd.DataFrame({"a": x, "id": np.arange(40.0)}), pd.DataFrame({"y": np.cos(x) + 0.1 * rng.normal(size=40)}),
        unit_col="id", categorical_vars=["id"])
awkward("an output that is an exact linear function of an input (noise goes to its floor)",
        pd.DataFrame({"a": x, "b": rng.normal(size=40)}), pd.DataFrame({"y": 2.0 * x + 1.0}))
awkward("outputs on wildly different scales, not standardised",
        pd.DataFrame({"a": x}), pd.DataFrame({"small": 1e-6 * np.sin(x), "big": 1e6 * np.sin(x) + 1e5 * rng.normal(size=40)}))
awkward("text categories", pd.DataFrame({"a": x, "g": rng.choice(["u", "v", "w"], 40)}),
        pd.DataFrame({"y": np.sin(x)}), categorical_vars=["g"])
# ---- edges of scale -------------------------------------------------------------------------------------------
def many_outputs():
    N, P = 60, 3000
    xs = rng.normal(size=(N, 2))
    Y = pd.DataFrame(np.sin(xs[:, :1] * rng.uniform(0.5, 2, P)) + 0.1 * rng.normal(size=(N, P)),
                     columns=[f"o{p}" for p in range(P)])
    g = GPSearch(X=pd.DataFrame(xs, columns=["a", "b"]), Y=Y)
    t0 = time.time()
    g.penalized_optimization(random_seed=1)
    return f"{P} outputs x N = {N}: {time.time() - t0:.1f} s, {len(g.models)} models"


def many_columns(D):
    N = 200
    xs = rng.normal(size=(N, D))
    g = GPSearch(X=pd.DataFrame(xs, columns=[f"c{d}" for d in range(D)]), Y=pd.DataFrame({"y": xs[:, 0] + np.sin(xs[:, 1])}))
    try:
        g.penalized_optimization(random_seed=1)
    except NotImplementedError as e:
        return "refused by name: " + str(e)[:90]
    return g.models["y"].kernel_name


def odd_size():
    N = 4097            # Npad = 4160: one 64-row granule past a multiple of 128
    xs = rng.uniform(-3, 3, size=(N, 2))
    g = GPSearch(X=pd.DataFrame(xs, columns=["a", "b"]),
                 Y=pd.DataFrame({"y": np.sin(xs[:, 0]) + 0.3 * xs[:, 1] + 0.1 * rng.normal(size=N)}))
    t0 = time.time()
    g.penalized_optimization(random_seed=1)
    m = g.models["y"]
    return f"N = {N}: {time.time() - t0:.1f} s, {m.kernel_name}, noise {float(m.likelihood.variance.value):.4f}"


def float32_mode():
    from waveome_amd.utilities import set_precision

    set_precision("float32")
    try:
        N = 700
        xs = rng.uniform(-3, 3, size=(N, 2))
        g = GPSearch(X=pd.DataFrame(xs, columns=["a", "b"]),
                     Y=pd.DataFrame({"y": np.sin(xs[:, 0]) + 0.3 * xs[:, 1] + 0.1 * rng.normal(size=N)}))
        g.penalized_optimization(random_seed=1)
        m = g.models["y"]
        m.get_feature_importances()
        return f"float32: {m.kernel_name}, noise {float(m.likelihood.variance.value):.4f}, importances {m.feature_importances}"
    finally:
        set_precision("float64")


def nan_refused():
    xs = rng.normal(size=(30, 1))
    y = np.sin(xs[:, 0]); y[3] = np.nan
    try:
        GPSearch(X=pd.DataFrame(xs, columns=["a"]), Y=pd.DataFrame({"y": y}))
    except AssertionError as e:
        return "refused like the reference: " + str(e)
    raise AssertionError("NaN in Y accepted")


step("scale: 3000 outputs", many_outputs)
step("scale: 30 input columns", lambda: many_columns(30))
step("scale: 40 input columns (limit 32)", lambda: many_columns(40))
step("scale: N = 4097", odd_size)
step("scale: float32 storage through set_precision", float32_mode)
step("input: NaN in Y", nan_refused)
print(f"# {len(failures)} failures" + ("" if not failures else ": " + "; ".join(failures)))
sys.exit(1 if failures else 0)
