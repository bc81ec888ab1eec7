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
from .utilities import (calc_bic, calc_feature_importance_components, calc_rsquare, find_variance_components,
                        print_kernel_names, search_through_kernel_list_)

_default_ctx: Optional[Context] = None


def default_context(device: int = 0) -> Context:
    global _default_ctx
    if _default_ctx is None or _default_ctx.device != device:
        _default_ctx = Context(device)
    return _default_ctx


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
        self.fit_result = None
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
        ctx = self._ctx or default_context()
        spec = self._spec()
        return spec, Problem(ctx, self.data[0], self.data[1], spec, max_batch=max_batch)

    def optimize_params(self, optimizer="scipy", data=None, num_opt_iter=50000, **_ignored):
        """L-BFGS-B on -(LML + log prior) (reference: model_classes.py:299-342, scipy branch)."""
        if optimizer not in ("scipy", None):
            raise NotImplementedError("only optimizer='scipy' (L-BFGS-B) is implemented on the GPU path")
        self.optimizer = "scipy"
        spec, prob = self._problem()
        res = prob.fit([0], spec.unconstrain()[None, :], lbfgs_opts(maxiter=num_opt_iter))
        spec.assign_theta(spec.constrain(res["u"][0]))
        self.fit_result = {k: v[0] for k, v in res.items()}
        prob.close()
        return None

    def maximum_log_likelihood_objective(self, data=None):
        spec, prob = self._problem()
        val, st = prob.lml([0], spec.theta()[None, :])
        prob.close()
        return float(val[0])

    def log_prior_density(self):
        return self._spec().log_prior()

    def log_posterior_density(self, data=None):
        return self.maximum_log_likelihood_objective(data) + self.log_prior_density()

    def training_loss(self, data=None):
        return -self.log_posterior_density(data)

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
        mean, var = prob.predict(0, spec.theta(), include_noise=False, Xnew=self._xnew(Xnew))
        prob.close()
        return mean[:, None], var[:, None]

    def predict_y(self, Xnew=None, full_cov=False, full_output_cov=False):
        """Posterior mean / variance of y (model_classes.py:536-544)."""
        if full_cov or full_output_cov:
            raise NotImplementedError("full_cov=True is not implemented")
        spec, prob = self._problem()
        mean, var = prob.predict(0, spec.theta(), include_noise=True, Xnew=self._xnew(Xnew))
        prob.close()
        return mean[:, None], var[:, None]

    def component_means(self, data=None, only=False):
        """[full model, without additive component 0, 1, ...] training means of the whitened SVGP the
        reference evaluates in calc_feature_importance_components (utilities.py:614-707); with `only`,
        [full, component 0 alone, 1 alone, ...] as individual_kernel_predictions does for calc_rsquare
        (utilities.py:162-172, 770-780)."""
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
        out = prob.whitened_means(0, spec.theta(), masks)
        prob.close()
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
        """Drop additive terms with variance < var_cutoff, then factors whose lengthscale is
        >= 3 x the input range (model_classes.py:1029-1079)."""
        var_parts = find_variance_components(self.kernel, sum_reduce=False)
        var_parts = np.asarray(var_parts).reshape(-1)
        var_flag = np.where(var_parts >= var_cutoff)[0]
        if len(var_flag) > 1:
            self.kernel = K.Sum([self.kernel.kernels[i] for i in var_flag])
        elif len(var_flag) == 1:
            if len(var_parts) > 1:
                self.kernel = self.kernel.kernels[var_flag[0]]
        else:
            self.kernel = K.Constant()
        if hasattr(self.kernel, "kernels"):
            X = self.data[0] if data is None else np.asarray(data[0])
            self.kernel = search_through_kernel_list_(kernel_list=self.kernel.kernels,
                                                      list_type=self.kernel.name, X=X)
        return None
