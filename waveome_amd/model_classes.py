"""PSVGP - the per-output model object `GPSearch.models[...]` holds.

Mirrors the attribute surface downstream waveome code reads from
waveome/model_classes.py::PSVGP (:1082-1126) - `.kernel`, `.likelihood.variance`,
`.mean_function.c`, `.kernel_name`, `.penalization_factor`, `.trainable_parameters`,
`.data`, `.optimize_params`, `.cut_kernel_components`, `.calc_metric`,
`.maximum_log_likelihood_objective`, `.log_posterior_density` - for the configuration this
path covers: Gaussian likelihood, inducing points = training inputs.  For that model the
optimal SVGP bound equals the exact log marginal likelihood, which is what the
GPU evaluates; q_mu / q_sqrt are therefore not optimisation variables here."""
from __future__ import annotations

import copy
from typing import Optional

import numpy as np

from . import kernels as K
from .engine import Context, Problem, lbfgs_opts
from .spec import ModelSpec
from .spec import _expand
from .utilities import (_gaussian_logdensity, restart_points, calc_bic, calc_feature_importance_components, calc_rsquare, find_variance_components,
                        print_kernel_names, search_through_kernel_list_)

_default_ctx: Optional[Context] = None


def default_context(device: Optional[int] = None) -> Context:
    """The context API calls use when none is passed: GPU 0, or under torch.distributed (one process per GPU) the
    GPU of this rank - LOCAL_RANK modulo the visible devices."""
    from . import _lib, parallel
    from .utilities import get_precision

    global _default_ctx
    if device is None:
        device = 0
        if parallel.dist_active():
            import ctypes

            n = ctypes.c_int32(0)
            _lib.load().wv_device_count(ctypes.byref(n))
            device = parallel.local_device_index(n.value)
    if _default_ctx is None or _default_ctx.device != device or _default_ctx.dtype != get_precision():
        _default_ctx = Context(device, dtype=get_precision())
    return _default_ctx


class ProblemPool:
    """Device problems on one (X, Y) keyed by kernel structure; `Y` holds every output column."""

    def __init__(self, ctx, X, Y, max_batch=8):  # 8 slots: the leave-one-component-out Gram matrices factor as a batch
        self.ctx, self.X, self.Y, self.max_batch = ctx, X, Y, max_batch
        self._probs = {}

    @staticmethod
    def signature(spec):
        d = spec.to_dict()
        for prm in d["params"]:  # values do not matter (theta is passed with every call), structure and flags do
            prm.pop("init", None)
        return repr(d)

    def get(self, spec):
        key = self.signature(spec)
        if key not in self._probs:
            self._probs[key] = Problem(self.ctx, self.X, self.Y, spec, max_batch=self.max_batch)
        return self._probs[key]

    def close(self):
        for prob in self._probs.values():
            prob.close()
        self._probs = {}


class _Inducing:
    """`model.inducing_variable.Z` of the reference's SVGP: here always the training inputs (exact GP, Z = X)."""

    def __init__(self, X):
        self.Z = K.Parameter(0.0, transform="identity", trainable=False, name="inducing_variable.Z")
        self.Z.value = np.asarray(X, dtype=np.float64)        # (a matrix, where kernel parameters are scalars)
        self.Z.numpy = lambda X=self.Z.value: X


class PSVGP:
    def __init__(self, X, Y, mean_function=None, kernel=None, verbose=False, num_latent_gps=1, dtype=None,
                 penalized_options={}, sparse_options={}, variational_options={}, context=None):
        if num_latent_gps != 1:
            raise NotImplementedError("multi-output models are outside the GPU path")
        lik = variational_options.get("likelihood", "gaussian")
        if lik != "gaussian":
            raise NotImplementedError(f"likelihood {lik!r}: only the Gaussian path is implemented")
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64).reshape(len(X), -1)
        assert np.isnan(X).sum() == 0, "Missing values in X found. This is currently not allowed!"
        assert np.isnan(Y).sum() == 0, "Missing values in Y found. This is currently not allowed!"
        self.data = (X, Y)
        self.mean_function = copy.deepcopy(mean_function) if mean_function is not None else K.ConstantMean(0.0)
        self.kernel = copy.deepcopy(kernel) if kernel is not None else K.SquaredExponential(active_dims=[0])
        self.likelihood = K.GaussianLikelihood()
        self.verbose = verbose
        self.optimizer = None
        self.kernel_name = ""
        self.feature_importances = None
        self.num_inducing_points = X.shape[0]  # exact GP: every training point (Z = X)
        self._ctx = context
        self._pool, self._col = None, 0  # shared device problems and this model's column in them (GPSearch)
        self.fit_result = None
        self.unit_col = None  # model_classes.py:814
        self.penalization_search_results = None
        self.set_penalization_factor(penalized_options.get("penalization_factor", 1.0))
        self.update_kernel_name()

    # ---- structure helpers ------------------------------------------------------
    def _spec(self) -> ModelSpec:
        return ModelSpec(self.kernel, self.data[0].shape[1], self.likelihood, self.mean_function)

    @property
    def parameters(self):
        return self._spec().param_objs

    @property
    def trainable_parameters(self):
        return [p for p in self.parameters if p.trainable]

    def update_kernel_name(self):
        try:
            self.kernel_name = print_kernel_names(self.kernel, with_idx=True)
        except TypeError:
            self.kernel_name = print_kernel_names(self.kernel)
        if type(self.kernel_name) not in [list, str]:
            self.kernel_name = "+".join(list(self.kernel_name))

    def set_penalization_factor(self, penalization_factor, use_prior=True):
        """Horseshoe(scale = 1/pf) on every kernel variance parameter (model_classes.py:837-864)."""
        self.penalization_factor = float(penalization_factor)
        if use_prior:
            prior = K.Horseshoe(1.0 / penalization_factor) if penalization_factor > 0 else None
            spec = self._spec()
            for name, p in zip(spec.param_names, spec.param_objs):
                if "kernel" in name and "variance" in name:
                    p.prior = prior

    # ---- GPU-backed numerics ----------------------------------------------------------
    def _problem(self, max_batch=2):
        """(spec, problem) for this model's data and current structure.  Models built by GPSearch share a
        ProblemPool: outputs whose (pruned) kernels have the same structure reuse one device problem that holds
        every output column, instead of allocating one each (7 + 2 ms per model, as much as a small fit).
        Prediction and the component means are float64 computations on any problem (a float32 one serves them on a
        float64 twin inside the library: include/waveome_amd.h, wv_predict)."""
        ctx = self._ctx or default_context()
        spec = self._spec()
        if self._pool is not None:
            return spec, self._pool.get(spec)
        return spec, Problem(ctx, self.data[0], self.data[1], spec, max_batch=max_batch)

    def _release(self, prob):
        if self._pool is None:
            prob.close()

    def optimize_params(self, optimizer="scipy", data=None, num_opt_iter=50000, **_ignored):
        """L-BFGS-B on -(LML + log prior) (reference: model_classes.py:299-342, scipy branch).  A fit that fails
        at its start point (no finite evaluation: K_y not positive definite there) is retried from randomised
        parameters, up to 5 times, as the reference's except-branch does (:323-340); if every attempt fails the
        parameters are left as they were and a warning names the outcome."""
        if optimizer not in ("scipy", None):
            raise NotImplementedError("only optimizer='scipy' (L-BFGS-B) is implemented on the GPU path")
        self._own_data_only(data, "optimize_params")
        self.optimizer = "scipy"
        spec, prob = self._problem()
        opts = lbfgs_opts(maxiter=num_opt_iter)
        start = spec.unconstrain()[None, :]
        try:
            res = prob.fit([self._col], start, opts)
            attempt = 0
            while not np.isfinite(res["loss"][0]) and attempt < 5:
                attempt += 1
                res = prob.fit([self._col], restart_points(spec.n_trainable, attempt)[-1:], opts)
        finally:
            self._release(prob)
        self.fit_result = {k: v[0] for k, v in res.items()}
        if np.isfinite(res["loss"][0]):
            spec.assign_theta(spec.constrain(res["u"][0]))
        else:
            import warnings

            warnings.warn("optimize_params: no finite objective value from the start point or 5 randomised "
                          "restarts; parameters left unchanged (status "
                          f"{int(res['status'][0])})")
        return None

    def _own_data_only(self, data, what):
        """The reference's models take `data` at every call (GPflow's SVGP holds none) and GPSearch always passes the
        training data; this mirror holds its data.  None or the training data itself is fine - anything else is
        refused instead of silently answering for the training data."""
        if data is None:
            return
        X = np.asarray(data[0], dtype=np.float64)
        Y = np.asarray(data[1], dtype=np.float64).reshape(len(X), -1)
        if X.shape != self.data[0].shape or Y.shape != self.data[1].shape or not (
                np.array_equal(X, self.data[0]) and np.array_equal(Y, self.data[1])):
            raise NotImplementedError(f"{what}: only the model's own training data is supported on the GPU path "
                                      "(build a PSVGP on the other data, or use predict_log_density for held-out rows)")

    def maximum_log_likelihood_objective(self, data=None, use_factor=False):
        """The exact LML (the bound of the reference's SVGP with Z = X at its optimal q).  use_factor: minus
        penalization_factor x the sum of the variance components - the L1-style penalty the reference keeps behind
        this switch and never turns on (model_classes.py:817-835)."""
        self._own_data_only(data, "maximum_log_likelihood_objective")
        spec, prob = self._problem()
        val, st = prob.lml([self._col], spec.theta()[None, :])
        self._release(prob)
        out = float(val[0])
        if use_factor:
            from .utilities import find_variance_components

            out -= self.penalization_factor * float(np.sum(find_variance_components(self.kernel)))
        return out

    def log_prior_density(self):
        return self._spec().log_prior()

    def log_posterior_density(self, data=None):
        return self.maximum_log_likelihood_objective(data) + self.log_prior_density()

    def training_loss(self, data=None):
        return -self.log_posterior_density(data)

    # ---- the SVGP surface code downstream of a fit reads off the reference's models --------------------
    @property
    def inducing_variable(self):
        """`.inducing_variable.Z`: the inducing points = the training inputs, not trainable (model_classes.py:99-100, 169)."""
        return _Inducing(self.data[0])

    def _whitened_state(self, want_sqrt):
        spec, prob = self._problem()
        try:
            return prob.whitened_state(self._col, spec.theta(), want_sqrt=want_sqrt)
        finally:
            self._release(prob)

    @property
    def q_mu(self):
        """[N, 1]: the whitened variational mean at its optimum for the current parameters (what the reference's
        optimiser drives its 125 k extra variables to): chol(K + 1e-6 I)' K_y^-1 (y - c).  Computed on request."""
        return self._whitened_state(False)[0][:, None]

    @property
    def q_sqrt(self):
        """[1, N, N] lower triangular: the Cholesky factor of the optimal whitened covariance
        (I + L'L / sigma^2)^-1 (gpflow's FillTriangular layout).  Two extra factorisations, computed on request."""
        return self._whitened_state(True)[1][None, :, :]

    # ---- prediction at the training inputs (the only Xnew this path uses) ----------------
    def _xnew(self, Xnew):
        """None / the training matrix itself -> the cheap training-input path; anything else is passed on."""
        if Xnew is None:
            return None
        Xnew = np.asarray(Xnew, dtype=np.float64)
        if Xnew.shape == self.data[0].shape and np.array_equal(Xnew, self.data[0]):
            return None
        return Xnew

    def predict_f(self, Xnew=None, full_cov=False, full_output_cov=False):
        """Posterior mean / variance of f (model_classes.py:526-534); full_cov is not supported."""
        if full_cov or full_output_cov:
            raise NotImplementedError("full_cov=True is not implemented")
        spec, prob = self._problem()
        mean, var = prob.predict(self._col, spec.theta(), include_noise=False, Xnew=self._xnew(Xnew))
        self._release(prob)
        return mean[:, None], var[:, None]

    def predict_y(self, Xnew=None, full_cov=False, full_output_cov=False):
        """Posterior mean / variance of y (model_classes.py:536-544)."""
        if full_cov or full_output_cov:
            raise NotImplementedError("full_cov=True is not implemented")
        spec, prob = self._problem()
        mean, var = prob.predict(self._col, spec.theta(), include_noise=True, Xnew=self._xnew(Xnew))
        self._release(prob)
        return mean[:, None], var[:, None]

    def predict_f_samples(self, Xnew=None, num_samples=None, full_cov=True, full_output_cov=False):
        """Posterior function draws - what the reference's plotting helpers call (predictions.py); they need the full
        posterior covariance, which this path does not form (section 8 of DESIGN.md: plotting is out of scope)."""
        raise NotImplementedError("predict_f_samples needs the full posterior covariance (full_cov=True), which the "
                                  "GPU path does not compute; predict_f / predict_y give means and variances")

    def predict_log_density(self, data):
        """log N(y | mean_f, var_f + sigma^2) per row (gpflow GPModel.predict_log_density, Gaussian likelihood)."""
        X, Y = data
        mean, var = self.predict_y(X)
        Y = np.asarray(Y, dtype=np.float64).reshape(-1, 1)
        return _gaussian_logdensity(Y, mean, var)[:, 0]

    # ---- restarts and the k-fold search over penalisation factors ----------------------------------
    def _fit_tasks(self, X, Y, factors, starts, num_opt_iter=50000):
        """Fit one task per row of `starts` (unconstrained start points) on data (X, Y[:, 0]) with
        penalisation factor factors[i]; tasks with a prior share one batched call with a per-task
        Horseshoe scale, factor-0 tasks (no prior, :844-861) share another."""
        ctx = self._ctx or default_context()
        factors = np.asarray(factors, dtype=np.float64)
        out = {k: np.zeros(len(factors)) for k in ("loss", "lml", "nit", "nfev", "status")}
        out["u"] = np.zeros_like(starts)
        for with_prior in (True, False):
            sel = np.nonzero((factors > 0) == with_prior)[0]
            if len(sel) == 0:
                continue
            twin = self._twin(X, Y, 1.0 if with_prior else 0.0)
            prob = Problem(ctx, X, Y, twin._spec(), max_batch=max(1, min(len(sel), 64)))
            try:
                res = prob.fit(np.zeros(len(sel), dtype=np.int32), starts[sel], lbfgs_opts(maxiter=num_opt_iter),
                               prior_scale=(1.0 / factors[sel]) if with_prior else None)
            finally:
                prob.close()
            for k in ("loss", "lml", "nit", "nfev", "status"):
                out[k][sel] = res[k]
            out["u"][sel] = res["u"]
        return out

    def _twin(self, X, Y, penalization_factor):
        """Same structure, flags and values on other data / another factor (the reference's deepcopy(self))."""
        twin = PSVGP(X, Y, mean_function=self.mean_function, kernel=self.kernel, verbose=self.verbose,
                     penalized_options={"penalization_factor": penalization_factor}, context=self._ctx)
        twin.likelihood = copy.deepcopy(self.likelihood)
        twin.unit_col = self.unit_col
        return twin

    def _restart_points(self, num_restart, loc=0.0, scale=1.0, random_seed=None):
        return restart_points(self._spec().n_trainable, num_restart, loc, scale, random_seed)

    def randomize_params(self, loc: float = 0.0, scale: float = 1.0, random_seed=None):
        """Assign every trainable parameter a value whose unconstrained coordinate is drawn from N(loc, scale)
        under `np.random.seed(random_seed)` (model_classes.py:181-234)."""
        np.random.seed(random_seed)
        spec = self._spec()
        u = np.random.normal(loc=loc, scale=scale, size=spec.n_trainable)
        spec.assign_theta(spec.constrain(u))
        return None

    def random_restart_optimize(self, data=None, num_restart=5, randomize_kwargs={}, optimize_kwargs={}):
        """`num_restart` randomised fits in one batched call; keeps the one with the highest
        `maximum_log_likelihood_objective` (no prior term, :511-519)."""
        X, Y = self.data if data is None else (np.asarray(data[0], dtype=np.float64),
                                               np.asarray(data[1], dtype=np.float64).reshape(len(data[0]), -1))
        starts = self._restart_points(num_restart, **randomize_kwargs)
        res = self._fit_tasks(X, Y, np.full(num_restart, self.penalization_factor), starts,
                              optimize_kwargs.get("num_opt_iter", 50000))
        lml = np.where(np.isfinite(res["lml"]), res["lml"], -np.inf)
        best = int(np.argmax(lml))
        spec = self._spec()
        spec.assign_theta(spec.constrain(res["u"][best]))
        self.fit_result = {k: v[best] for k, v in res.items()}
        self.optimizer = "scipy"
        return None

    def penalization_search(self, data=None, penalization_factor_list=[0.0, 1.0, 10.0, 100.0], k_fold=3,
                            fit_best=True, max_jobs=-1, show_progress=True, parallel_object=None,
                            randomization_options={}, optimization_options={}, random_seed=None, num_restart=5,
                            selection_type="se"):
        """k-fold search over penalisation factors (model_classes.py:866-998): every (factor, fold) pair
        is `num_restart` randomised fits on the training rows, scored by the mean holdout
        `predict_log_density`; the factor with the best mean (minus one standard error for
        selection_type="se") wins and, with fit_best, is refitted on all the data.

        All factors x restarts of one fold are one batched GPU call (same training matrix).  The
        reference's `parallel_fit` deep-copies the model but never applies `pf` to the copy (:900-918), so
        there every factor is fitted under the model's current prior; here the factor is applied, which is
        what the surrounding selection logic assumes.  `max_jobs`, `show_progress`, `parallel_object` are
        accepted and ignored."""
        from .regularization import make_folds

        data = self.data if data is None else data
        X = np.asarray(data[0], dtype=np.float64)
        Y = np.asarray(data[1], dtype=np.float64).reshape(len(X), -1)
        folds = make_folds(X, self.unit_col, k_fold, random_seed)
        rand = dict(randomization_options)
        rand.setdefault("random_seed", random_seed)
        factors = [float(f) for f in penalization_factor_list]
        starts = self._restart_points(num_restart, **rand)
        n_opt = optimization_options.get("num_opt_iter", 50000)
        results = []
        spec = self._spec()
        for h, fold in enumerate(folds):
            Xtr, Ytr = np.delete(X, fold, axis=0), np.delete(Y, fold, axis=0)
            task_f = np.repeat(factors, num_restart)
            res = self._fit_tasks(Xtr, Ytr, task_f, np.tile(starts, (len(factors), 1)), n_opt)
            for i, pf in enumerate(factors):
                blk = slice(i * num_restart, (i + 1) * num_restart)
                lml = np.where(np.isfinite(res["lml"][blk]), res["lml"][blk], -np.inf)
                u = res["u"][blk][int(np.argmax(lml))]
                twin = self._twin(Xtr, Ytr, pf)
                tspec = twin._spec()
                tspec.assign_theta(tspec.constrain(u))
                holdout = np.mean(twin.predict_log_density((X[fold], Y[fold])))
                results.append([pf, h, holdout])
        table = np.array(results)  # rows (factor, fold, mean holdout log-density)
        self.penalization_search_results = table
        # per factor: mean over folds, minus one standard error for selection_type "se"; the first factor with the
        # strictly highest score wins; a search without a finite score falls back to no penalisation (:959-996)
        by_factor = np.array([table[table[:, 0] == f, 2] for f in factors])
        score = by_factor.mean(axis=1)
        if selection_type == "se":
            score = score - by_factor.std(axis=1) / np.sqrt(k_fold)
        score = np.where(np.isnan(score), -np.inf, score)
        if np.all(np.isneginf(score)):
            if self.verbose:
                print("Search error, returning no penalization")
            best_factor = 0.0
        else:
            best_factor = factors[int(np.argmax(score))]
        if self.verbose:
            print(f"Best penalization factor found from search: {best_factor}")
        if fit_best:
            self.set_penalization_factor(best_factor)
            self.random_restart_optimize(data=(X, Y), randomize_kwargs=rand, optimize_kwargs=optimization_options,
                                         num_restart=num_restart)
        return None

    def component_means(self, data=None, only=False):
        """[full model, without additive component 0, 1, ...] training means of the whitened SVGP the
        reference evaluates in calc_feature_importance_components (utilities.py:614-707); with `only`,
        [full, component 0 alone, 1 alone, ...] as individual_kernel_predictions does for calc_rsquare
        (utilities.py:162-172, 770-780)."""
        self._own_data_only(data, "component_means")
        spec, prob = self._problem()
        comps = self.kernel.kernels if self.kernel.name == "sum" else [self.kernel]
        # additive component k owns a contiguous run of flattened terms (products of sums expand)
        counts = [len(_expand(k)) for k in comps]
        starts = np.concatenate([[0], np.cumsum(counts)])
        n_terms = int(starts[-1])
        full = (1 << n_terms) - 1
        masks = [full]
        if self.kernel.name == "sum":
            for k in range(len(comps)):
                own = ((1 << counts[k]) - 1) << int(starts[k])
                masks.append(own if only else full & ~own)
        out = prob.whitened_means(self._col, spec.theta(), masks)
        self._release(prob)
        return out

    def get_feature_importances(self, data=None, return_value="log_bf"):
        """model_classes.py:546-573."""
        data = self.data if data is None else data
        self.feature_importances = list(calc_feature_importance_components(model=self, data=data,
                                                                           return_value=return_value))
        return None

    def calc_metric(self, data=None, metric="BIC"):
        assert metric == "BIC", "Only BIC currently allowed."
        # the reference counts Parameter objects, q_mu and q_sqrt included (utilities.py:94-95,
        # model_search.py:2313-2319): add those two so BIC values are comparable
        return calc_bic(loglik=self.log_posterior_density(data), n=self.data[0].shape[0],
                        k=len(self.trainable_parameters) + 2)

    def cut_kernel_components(self, data=None, var_cutoff: float = 0.1):
        """Post-fit pruning (model_classes.py:1029-1079): additive components whose variance (a product's: the
        product of its factors') is below `var_cutoff` are dropped - none left makes the kernel a Constant(), one
        left stands alone - and what remains of a Sum / Product is passed through the lengthscale rule
        (`search_through_kernel_list_`: lengthscale >= 3 x the input range goes)."""
        components = self.kernel.kernels if self.kernel.name == "sum" else [self.kernel]
        strength = np.asarray(find_variance_components(self.kernel, sum_reduce=False), dtype=np.float64).reshape(-1)
        kept = [c for c, v in zip(components, strength) if v >= var_cutoff]
        if not kept:
            self.kernel = K.Constant()
        elif self.kernel.name == "sum":
            self.kernel = kept[0] if len(kept) == 1 else K.Sum(kept)
        if hasattr(self.kernel, "kernels"):
            X = self.data[0] if data is None else np.asarray(data[0])
            self.kernel = search_through_kernel_list_(kernel_list=self.kernel.kernels, list_type=self.kernel.name,
                                                      X=X)
        return None
