"""GPSearch - the caller-facing entry of the path: constructor preprocessing and
`penalized_optimization()` (reference waveome/model_search.py:47-517), GPflow/Ray-free.

What is kept verbatim: argument names and defaults, categorical factorisation, float casts, NaN
asserts, ddof=1 standardisation, `self.models[out_name]`, `self.model_selection_type`,
`self.run_parameters`, the progress prints.  What is replaced: the per-output Ray task
(:250-393) becomes one batched GPU fit over all outputs (sharded over ranks when
torch.distributed is initialised); `time.sleep(10)` and `rm -rf /tmp/ray` are not reproduced."""
from __future__ import annotations

import copy
import time
import warnings

import numpy as np
import pandas as pd

from . import kernels as K
from . import parallel
from .engine import Problem, lbfgs_opts
from .model_classes import PSVGP, ProblemPool, default_context
from .regularization import full_kernel_build
from .spec import ModelSpec


class GPSearch:
    """Gaussian process model search class (constructor: model_search.py:73-191)."""

    def __init__(self, X, Y, unit_col=None, standardize_X=True, Y_transform=None, categorical_vars=[],
                 outcome_likelihood="gaussian"):
        for frame, label in ((X, "X"), (Y, "Y")):
            if not isinstance(frame, pd.DataFrame):
                raise TypeError(f"{label} is not a Pandas DataFrame")
        # the unit column is a categorical column too (the reference appends it to its mutable default list)
        cat_names = list(categorical_vars) + ([unit_col] if unit_col is not None and unit_col not in categorical_vars
                                              else [])
        X, self.categorical_dict = self._encode_categoricals(X.copy(), cat_names)
        X, Y = self._all_float(X, "X"), self._all_float(Y, "Y")
        self.X, self.Y = X.copy(), Y.copy()
        self.feat_names, self.out_names = list(X.columns), list(Y.columns)
        self.likelihood = outcome_likelihood
        self.cat_idx = [self.feat_names.index(name) for name in cat_names]
        self.unit_idx = None if unit_col is None else self.feat_names.index(unit_col)
        self.cont_idx = [i for i in range(X.shape[1]) if i not in set(self.cat_idx)]
        if standardize_X:
            self._standardize_continuous()
        self._transform_outputs(Y_transform)
        self.rank_stats = None   # per-rank busy times of the last sharded job (torch.distributed only)

    def _block_sort_column(self):
        """Column the device-side copies of the data are sorted by (rows of a unit / category contiguous): candidate
        structures in which every term carries that column's Categorical factor are block diagonal, and the device
        products then skip the zero blocks.  The unit column if there is one, else the first
        categorical column, else None."""
        if self.unit_idx is not None:
            return self.unit_idx
        return self.cat_idx[0] if self.cat_idx else None

    @staticmethod
    def _encode_categoricals(X, cat_names):
        """Text-like categorical columns become float codes in order of first appearance (pd.factorize); the
        (codes, uniques) pair is kept per column for decoding (model_search.py:98-104)."""
        encodings = {}
        for name in cat_names:
            if str(X[name].dtype) in ("object", "string", "category", "str"):
                print(f"Converting {name} to numeric")
                encodings[name] = pd.factorize(X[name])
                X[name] = encodings[name][0].astype(float)
        return X, encodings

    @staticmethod
    def _all_float(frame, label):
        """Cast to float, refuse missing values (model_search.py:106-140)."""
        try:
            frame = frame.astype(float)
        except Exception:
            raise TypeError(f"{label} columns must all be float type. Perhaps use pandas.factorize() and"
                            " pandas.DataFrame.astype().")
        assert frame.isna().sum().sum() == 0, f"NAs in {label}, waveome cannot currently handle missing values!"
        return frame

    def _standardize_continuous(self):
        """Continuous columns to zero mean / unit sample standard deviation (pandas .std(): ddof = 1), categorical
        codes untouched; the originals and the moments are kept (model_search.py:155-168)."""
        cont = [self.feat_names[i] for i in self.cont_idx]
        self.X_original = self.X.copy()
        self.X_means, self.X_stds = self.X[cont].mean(axis=0), self.X[cont].std(axis=0)
        self.X[cont] = (self.X[cont] - self.X_means) / self.X_stds

    def _transform_outputs(self, how):
        """Y_transform: "standardize" (centre and scale) or "scale" (scale only) (model_search.py:172-191)."""
        if how not in ("standardize", "scale"):
            return
        if how == "standardize" and self.likelihood != "gaussian":
            warnings.warn("Standardizing Y without a gaussian likelihood is not advised!")
        self.Y_original = self.Y.copy()
        self.Y_stds = self.Y.std(axis=0)
        if how == "standardize":
            self.Y_means = self.Y.mean(axis=0)
            self.Y = self.Y - self.Y_means
        self.Y = self.Y / self.Y_stds

    def reverse_transform(self, array, feature_name=None, input_type="X", round_digits=1):
        """Values on the standardised scale back on the caller's scale (model_search.py:1677-1715): x * std + mean with
        the moments `_standardize_continuous` / `_transform_outputs` kept - for one column (`feature_name`) or, with
        None, all standardised columns at once (the array's last axis) - rounded to `round_digits` decimals."""
        if input_type not in ("X", "Y"):
            raise ValueError("Unknown type requested for transform!")
        if input_type == "X":
            assert hasattr(self, "X_stds"), "Standardize_X wasn't called in GPSearch()"
            scale, shift = self.X_stds, self.X_means
        else:
            assert hasattr(self, "Y_stds"), "Y_transform wasn't called in GPSearch()"
            scale = self.Y_stds
            shift = getattr(self, "Y_means", None)   # Y_transform="scale" keeps no means
        pick = (lambda s_: s_.to_numpy()) if feature_name is None else (lambda s_: s_[feature_name])
        offset = pick(shift) if shift is not None else 0.0
        return np.round(pick(scale) * np.asarray(array) + offset, decimals=round_digits)

    def run_penalized_search(self, *args, **kwargs):
        """Deprecated in the reference (model_search.py:933-959): it raises, and so does this."""
        raise NotImplementedError("run_penalized_search is deprecated, use penalized_optimization instead.")

    # ------------------------------------------------------------------------------------------
    def penalized_optimization(self, full_kernel=None, num_jobs=-1, verbose=False, mean_function=None,
                               kernel_options={"second_order_numeric": False,
                                               "categorical_numeric_interactions": True,
                                               "unit_numeric_interactions": False, "kerns": None},
                               penalization_factor=1.0, num_factor_iter=5, num_restart=0, sparse_options={},
                               variational_options={}, optimization_options={"optimizer": "scipy"},
                               random_seed=None, ray_dashboard=False, ray_logging=False, max_batch=None,
                               context=None):
        """Fit the saturated penalised model to every output column (model_search.py:197-517).

        Extra keyword arguments (not in the reference): `max_batch` (fits in flight per GPU) and
        `context`.  `num_jobs`, `ray_dashboard`, `ray_logging` are accepted and ignored."""
        self.model_selection_type = "penalized"
        if not hasattr(self, "run_parameters"):
            self.run_parameters = {}
        args = dict(locals())
        del args["self"]
        self.run_parameters["penalized_optimization"] = args
        if self.likelihood != "gaussian":
            raise NotImplementedError("only outcome_likelihood='gaussian' is implemented on the GPU path")
        if optimization_options.get("optimizer", "scipy") != "scipy":
            raise NotImplementedError("only optimization_options={'optimizer': 'scipy'} is implemented")
        if sparse_options:
            warnings.warn("sparse_options are ignored: the GPU path is the exact GP (all N points)")
        if random_seed is not None:
            np.random.seed(random_seed)
        if mean_function is None:
            mean_function = K.ConstantMean()
        if full_kernel is None:
            ko = dict(kernel_options)
            if ko.get("kerns") is None:
                ko["kerns"] = [K.SquaredExponential()]
            full_kernel, _ = full_kernel_build(cat_vars=self.cat_idx, num_vars=self.cont_idx,
                                               unit_idx=self.unit_idx, var_names=self.feat_names,
                                               return_sum=True, **ko)
        else:
            full_kernel = copy.deepcopy(full_kernel)
        variational_options = dict(variational_options, likelihood=self.likelihood)

        # rank 0's X and Y go out to every rank (RCCL broadcast under nccl; the reference's ray.put, :434-472)
        shared = parallel.broadcast_inputs(self.X.to_numpy(dtype=np.float64), self.Y.to_numpy(dtype=np.float64))
        Xn, Yn = shared.X, shared.Y
        P = Yn.shape[1]
        # penalization_factor=None: analytic factor 2 * 1.1 * sigma_hat * sqrt(N) * Phi^-1(1 - 0.1 / (2 p)) per
        # output, then up to num_factor_iter re-fits with sigma_hat = sqrt(mean predict_y variance)
        # (model_search.py:273-299, 332-375)
        iterating = penalization_factor is None
        if iterating:
            import scipy.stats

            from .utilities import find_variance_components

            if num_restart > 0:
                raise NotImplementedError("penalization_factor=None with num_restart > 0")
            num_params = len(np.asarray(find_variance_components(full_kernel, sum_reduce=False)).reshape(-1))
            zq = scipy.stats.norm().ppf(1 - (0.1 / (2 * num_params)))
            sigma_hat = np.ones(P) if num_factor_iter == 0 else np.std(Yn, axis=0)  # np.std(Series): ddof = 0
            pf = 2 * 1.1 * sigma_hat * np.sqrt(Xn.shape[0]) * zq
            if verbose:
                print("Setting penalization factor to", f" {pf}")
        else:
            pf = np.full(P, float(penalization_factor))
        # template model: structure, defaults and priors shared by every output
        template = PSVGP(Xn, Yn[:, :1], mean_function=mean_function, kernel=full_kernel, verbose=verbose,
                         penalized_options={"penalization_factor": float(pf[0])},
                         variational_options=variational_options, context=context)
        spec = template._spec()
        u0 = spec.unconstrain()
        n_starts = max(1, int(num_restart))
        starts = np.tile(u0, (P * n_starts, 1))
        if num_restart > 0:  # the same seeded N(0, 1) points for every output (model_classes.py:181-234, 498-503)
            from .utilities import restart_points

            starts = np.tile(restart_points(len(u0), n_starts, random_seed=random_seed), (P, 1))
        cols = np.repeat(np.arange(P, dtype=np.int32), n_starts)

        print(f"Building {len(self.out_names)} models...")
        start_time = time.time()
        ctx = context or default_context()
        rank, world = parallel.dist_info()
        nt = spec.n_trainable
        opts = lbfgs_opts(maxiter=optimization_options.get("num_opt_iter", 50000))

        has_prior = bool(np.all(pf > 0))
        if not has_prior and np.any(pf > 0):
            raise ValueError("penalization factors must be all positive or all zero")

        def fit_shard(ids, dealer=None):
            """The fits of tasks `ids` (single process / static shard), or - with `dealer` - of whatever chunks this
            rank draws from the job's dealer (parallel.dynamic_map); then the factor iteration on those tasks."""
            if dealer is None:
                mb = max_batch or min(len(ids), _default_batch(len(Xn)))
                prob = shared.problem(ctx, spec, max(1, mb))
                scale = (1.0 / pf[cols[ids]]) if has_prior else None
                res = prob.fit(cols[ids], starts[ids], opts, prior_scale=scale)
            else:
                mb = max_batch or min(-(-len(cols) // world) + dyn_chunk, _default_batch(len(Xn)))
                prob = shared.problem(ctx, spec, max(1, mb))
                ids, res = prob.fit_stream(dealer.pull, lambda t: (cols[t], None, starts[t],
                                                                   (1.0 / pf[cols[t]]) if has_prior else None),
                                           opts, first_fill=dealer.first_fill)
            u, pf_used = res["u"], pf[cols[ids]].copy()
            extra = np.zeros((len(ids), 2))  # total iterations / evaluations over the factor re-fits
            if iterating:
                live = np.arange(len(ids))
                for _ in range(int(num_factor_iter)):
                    keep = []
                    for k in live:
                        _, var = prob.predict(int(cols[ids][k]), spec.constrain(u[k]), include_noise=True)
                        new_pf = 2 * 1.1 * np.sqrt(np.mean(var)) * np.sqrt(Xn.shape[0]) * zq
                        if verbose:
                            print("New penalization factor:" f" {new_pf}")
                        if abs(new_pf - pf_used[k]) <= 1e-3 or new_pf > pf_used[k]:
                            continue  # converged, or a larger factor: keep the current fit (:354-364)
                        pf_used[k] = new_pf
                        keep.append(k)
                    live = np.array(keep, dtype=np.int64)
                    if len(live) == 0:
                        break
                    r2 = prob.fit(cols[ids][live], u[live], opts, prior_scale=1.0 / pf_used[live])
                    u[live] = r2["u"]
                    for key in ("loss", "lml", "status"):
                        res[key][live] = r2[key]
                    extra[live, 0] += r2["nit"]
                    extra[live, 1] += r2["nfev"]
            prob.close()
            rec = np.column_stack([res["loss"], res["lml"], res["nit"] + extra[:, 0], res["nfev"] + extra[:, 1],
                                   res["status"], pf_used, u])
            return rec if dealer is None else (ids, rec)

        # Under torch.distributed the outputs are dealt to the ranks in chunks, on demand (these fits run to SciPy's
        # tolerances: their iteration counts differ several-fold, a static split would wait for the unluckiest rank);
        # per-rank busy times -> self.rank_stats (max / mean rank time)
        dyn_chunk = max(1, min(_default_batch(len(Xn)) // 16, -(-len(cols) // (8 * world))))
        if parallel.dist_active():
            rec = parallel.dynamic_map(len(cols), lambda dealer: fit_shard(None, dealer), dyn_chunk, 6 + nt)
            self.rank_stats = parallel.last_map_stats()
            if verbose and rank == 0:
                print("rank_time_max_over_mean:", round(self.rank_stats["rank_time_max_over_mean"], 3),
                      "rank seconds:", [round(v, 2) for v in self.rank_stats["rank_seconds"]])
        else:
            rec = fit_shard(np.arange(len(cols)))
        self.fit_records = rec
        self.models = {}
        pool = ProblemPool(ctx, Xn, Yn)  # post-fit steps of outputs with the same pruned structure share a problem
        for p, feat in enumerate(self.out_names):
            block = rec[p * n_starts:(p + 1) * n_starts]
            # restarts: keep the highest maximum_log_likelihood_objective (model_classes.py:511-519)
            best = block[np.argmax(np.where(np.isfinite(block[:, 1]), block[:, 1], -np.inf))]
            if not np.isfinite(best[0]):  # the reference's task would have raised after its retries (:323-340)
                warnings.warn(f"penalized_optimization: no finite objective value for output {feat!r} "
                              f"(status {int(best[4])}); its model keeps the default parameter values")
                best = best.copy()
                best[6:6 + nt] = u0
            mod = PSVGP(Xn, Yn[:, p:p + 1], mean_function=mean_function, kernel=full_kernel, verbose=verbose,
                        penalized_options={"penalization_factor": float(best[5])},
                        variational_options=variational_options, context=context)
            mspec = mod._spec()
            mspec.assign_theta(mspec.constrain(best[6:6 + nt]))
            mod.fit_result = {"loss": best[0], "lml": best[1], "nit": int(best[2]), "nfev": int(best[3]),
                              "status": int(best[4])}
            mod.optimizer = "scipy"
            mod.cut_kernel_components(data=(Xn, Yn[:, p:p + 1]))
            mod.update_kernel_name()
            mod._pool, mod._col = pool, p
            mod.get_feature_importances(data=(Xn, Yn[:, p:p + 1]))  # model_search.py:383-387
            mod._pool, mod._col = None, 0  # the returned model stands on its own data again
            self.models[feat] = mod
        pool.close()
        c = len(self.out_names)
        elapsed_time = np.round((time.time() - start_time) / 60, 1)
        print(f"Finished {c} models (100%),", f"elapsed time: {elapsed_time} minutes")
        return None


    # ------------------------------------------------------------------------------------------
    def kernel_sweep(self, kernels, mean_function=None, penalization_factor=0.0, num_restart=0,
                     random_seed=None, num_opt_iter=50000, max_batch=None, context=None, keep_models=True):
        """Fit every (output column x candidate kernel structure) pair and tabulate the information
        criterion - the "Y-columns x candidate kernel structures" sweep of the north star (config C4:
        512 outputs x 6 kernels).  Each pair is what the reference's `kernel_test` evaluates for one
        candidate (model_search.py:2239-2334: PSVGP with penalization_factor = 0, then
        `calc_bic(log_posterior_density, n, k)`), except that the fit is the L-BFGS-B path of this build
        (the reference's kernel_test defaults to its Adam / natural-gradient branch, which is out of scope).

        kernels: dict name -> kernel (or a list; names default to the printed structure).
        Tasks are sharded over ranks when torch.distributed is initialised (cost ~ parameters per
        structure); a rank's tasks run as one multi-structure work queue (`wv_fit_multi`).  Returns a DataFrame indexed by
        (output, structure) with loss, lml, bic, nit, nfev, status; `self.sweep_models[(output, name)]`
        holds the fitted PSVGP objects when keep_models."""
        from .utilities import print_kernel_names

        if not isinstance(kernels, dict):
            kernels = {str(print_kernel_names(k, with_idx=True)): k for k in kernels}
        names = list(kernels)
        if mean_function is None:
            mean_function = K.ConstantMean()
        shared = parallel.broadcast_inputs(self.X.to_numpy(dtype=np.float64), self.Y.to_numpy(dtype=np.float64))
        Xn, Yn = shared.X, shared.Y  # (host-side models keep the caller's row order)
        shared = shared.sorted_by(self._block_sort_column())  # the device fits do not care, block structures gain
        N, P, S = Xn.shape[0], Yn.shape[1], len(names)
        ctx = context or default_context()
        opts = lbfgs_opts(maxiter=num_opt_iter)
        from .utilities import restart_points

        n_starts = max(1, int(num_restart))
        templates, specs = [], []
        for nm in names:
            t = PSVGP(Xn, Yn[:, :1], mean_function=mean_function, kernel=kernels[nm],
                      penalized_options={"penalization_factor": penalization_factor}, context=ctx)
            templates.append(t)
            specs.append(t._spec())
        width = max(sp.n_trainable for sp in specs)
        # task list: structure-major so that a rank's tasks of one structure share a Problem
        task_s = np.repeat(np.arange(S), P * n_starts)
        task_p = np.tile(np.repeat(np.arange(P), n_starts), S)
        starts = []
        for sp in specs:
            u0 = sp.unconstrain()
            blk = np.tile(u0, (P * n_starts, 1))
            if num_restart > 0:
                blk = np.tile(restart_points(len(u0), n_starts, random_seed=random_seed), (P, 1))
            starts.append(blk)
        costs = np.array([specs[s_].n_trainable for s_ in task_s], dtype=np.float64)

        def fit_shard(ids):
            # one multi-structure problem and one work queue for the whole shard (wv_fit_multi)
            out = np.full((len(ids), 5 + width), np.nan)
            used = np.unique(task_s[ids])
            local_of = {int(s_): k for k, s_ in enumerate(used)}
            sidx = np.array([local_of[int(s_)] for s_ in task_s[ids]], dtype=np.int32)
            U0 = np.zeros((len(ids), width))
            for row, t in enumerate(ids):
                s_ = task_s[t]
                U0[row, :specs[s_].n_trainable] = starts[s_][t - s_ * P * n_starts]
            mb = max_batch or min(len(ids), _default_batch(N))
            prob = shared.problem(ctx, [specs[s_] for s_ in used], max(1, mb))
            try:
                res = prob.fit_multi(task_p[ids].astype(np.int32), sidx, U0, opts)
            finally:
                prob.close()
            out[:, :5] = np.column_stack([res["loss"], res["lml"], res["nit"], res["nfev"], res["status"]])
            out[:, 5:5 + width] = res["u"][:, :width]
            return out

        rank, world = parallel.dist_info()
        n_tasks = len(task_s)

        def fit_stream(dealer):
            # this rank's share of a job dealt in chunks (parallel.dynamic_map): one problem holding every structure,
            # one work queue kept full across the chunks it draws
            mb = max_batch or min(-(-n_tasks // world) + dyn_chunk, _default_batch(N))
            prob = shared.problem(ctx, specs, max(1, mb))

            def task_args(t):
                U0 = np.zeros((len(t), width))
                for row, tt in enumerate(t):
                    s_ = task_s[tt]
                    U0[row, :specs[s_].n_trainable] = starts[s_][tt - s_ * P * n_starts]
                return task_p[t].astype(np.int32), task_s[t].astype(np.int32), U0, None

            try:
                ids, res = prob.fit_stream(dealer.pull, task_args, opts, width=width, first_fill=dealer.first_fill)
            finally:
                prob.close()
            out = np.full((len(ids), 5 + width), np.nan)
            out[:, :5] = np.column_stack([res["loss"], res["lml"], res["nit"], res["nfev"], res["status"]])
            out[:, 5:5 + width] = res["u"][:, :width]
            return ids, out

        # fits run to SciPy's tolerances differ several-fold in iterations (C4: 58 for Lin, 395 for Periodic per fit
        # on average, and as much again between columns): under torch.distributed the tasks are dealt to the ranks in
        # chunks, on demand, heaviest structures first (waveome/model_search.py:474-489 - Ray did this per task)
        dyn_chunk = max(1, min(_default_batch(N) // 16, -(-n_tasks // (8 * world))))
        if parallel.dist_active():
            rec = parallel.dynamic_map(n_tasks, fit_stream, dyn_chunk, 5 + width, costs=costs)
            self.rank_stats = parallel.last_map_stats()
        else:
            rec = fit_shard(np.arange(n_tasks))
        rows, self.sweep_models = [], {}
        for s_, nm in enumerate(names):
            sp = specs[s_]
            k_count = sum(1 for q in sp.param_objs if q.trainable) + 2  # + q_mu, q_sqrt (utilities.py:94-95)
            for p, feat in enumerate(self.out_names):
                blk = rec[(s_ * P + p) * n_starts:(s_ * P + p + 1) * n_starts]
                best = blk[np.argmax(np.where(np.isfinite(blk[:, 1]), blk[:, 1], -np.inf))]
                loglik = -best[0]  # log posterior density = LML + log prior = -training loss
                rows.append({"output": feat, "structure": nm, "loss": best[0], "lml": best[1],
                             "bic": round(2 * k_count - 2 * loglik, 2), "nit": int(best[2]), "nfev": int(best[3]),
                             "status": int(best[4])})
                if keep_models:
                    mod = PSVGP(Xn, Yn[:, p:p + 1], mean_function=mean_function, kernel=kernels[nm],
                                penalized_options={"penalization_factor": penalization_factor}, context=ctx)
                    ms_ = mod._spec()
                    ms_.assign_theta(ms_.constrain(best[5:5 + sp.n_trainable]))
                    mod.fit_result = rows[-1]
                    self.sweep_models[(feat, nm)] = mod
        self.sweep_results = pd.DataFrame(rows).set_index(["output", "structure"])
        return self.sweep_results


    # ------------------------------------------------------------------------------------------
    def run_search(self, kernels=None, max_depth=5, early_stopping=True, prune=True, keep_all=False,
                   metric_diff=6, num_restart=1, random_seed=None, num_jobs=-1, verbose=False, debug=False,
                   keep_only_best=True, max_batch=None, context=None, scorer=None):
        """Stepwise compositional kernel search for every output column (model_search.py:1069-1250 ->
        `full_kernel_search` :2987-3273).  Default `kernels` are the reference's: SquaredExponential,
        Matern12, Lin, Periodic(SquaredExponential).

        The candidates of all output columns at one search step are pooled and fitted in batched GPU
        calls (waveome_amd/kernel_search.py); `num_jobs` is accepted and ignored.  `num_restart > 1`: every
        candidate is fitted from that many seeded random start points and the fit with the best
        `maximum_log_likelihood_objective` is scored (`kernel_test` -> `random_restart_optimize`, :2283-2288).  `scorer` replaces the GPU scorer
        (test seam: tests/ drive the search logic with the CPU oracle).  Sets `self.models[out]` (the
        best model, fitted), `self.search_info[out]` = {"models", "edges", "best_model", "var_exp"}."""
        from . import kernel_search as ks
        from .utilities import calc_rsquare

        self.model_selection_type = "stepwise"
        self.verbose = verbose
        if self.likelihood != "gaussian":
            raise NotImplementedError("only outcome_likelihood='gaussian' is implemented on the GPU path")
        if kernels is None:
            kernels = [K.SquaredExponential(), K.Matern12(), K.Lin(), K.Periodic(K.SquaredExponential())]
        if random_seed is not None:
            np.random.seed(random_seed)
        shared = parallel.broadcast_inputs(self.X.to_numpy(dtype=np.float64), self.Y.to_numpy(dtype=np.float64))
        Xn, Yn = shared.X, shared.Y  # (host-side models keep the caller's row order)
        shared = shared.sorted_by(self._block_sort_column())  # the device fits do not care, block structures gain
        P = Yn.shape[1]
        print(f"Building {len(self.out_names)} models...")
        start_time = time.time()
        if scorer is None:
            scorer = ks.GpuScorer(Xn, Yn, context=context, max_batch=max_batch, num_restart=num_restart,
                                  random_seed=random_seed, shared=shared)
        # one process, the default scorer: the columns' searches run unsynchronised on one growing work queue
        # (WV_SEARCH_ASYNC=0: level by level); under torch.distributed the levels are sharded over the ranks
        import os

        use_async = (scorer.__class__ is ks.GpuScorer and scorer.mode == "multi"
                     and os.environ.get("WV_SEARCH_ASYNC", "1") != "0")
        kw = dict(cat_vars=self.cat_idx, verbose=verbose, max_depth=max_depth, keep_all=keep_all,
                  metric_diff=metric_diff, early_stopping=early_stopping, prune=prune, keep_only_best=keep_only_best)
        if use_async and parallel.dist_active():
            # every column's search lives on one rank, columns are dealt on demand, results gathered at the end
            out = ks.run_searches_sharded(P, Xn.shape[1], kernels, scorer, **kw)
            self.rank_stats = parallel.last_map_stats()
        elif use_async:
            out = ks.run_searches_async(range(P), Xn.shape[1], kernels, scorer, **kw)
        else:  # level by level; under torch.distributed each level's candidates are dealt to the ranks
            out = ks.run_searches(range(P), Xn.shape[1], kernels, scorer, **kw)
        self.models, self.search_info = {}, {}
        for p, feat in enumerate(self.out_names):
            res = out[p]
            best = res["models"][res["best_model"]]
            mod = PSVGP(Xn, Yn[:, p:p + 1], mean_function=K.ConstantMean(), kernel=best["kernel"], verbose=verbose,
                        penalized_options={"penalization_factor": 0.0}, context=context)
            mspec = mod._spec()
            u = np.asarray(best["fit"].get("u", []), dtype=np.float64)
            if len(u) == mspec.n_trainable and len(u):
                mspec.assign_theta(mspec.constrain(u))
            mod.fit_result = {k_: v for k_, v in best["fit"].items() if k_ != "u"}
            mod.optimizer = "scipy"
            mod.kernel_name = res["best_model"]
            best["model"] = mod
            if scorer.__class__ is ks.GpuScorer:
                res["var_exp"] = calc_rsquare(mod)
            self.models[feat] = mod
            self.search_info[feat] = res
        totals = getattr(scorer, "job_totals", None) or {}
        self.search_stats = {"fits": totals.get("fits", getattr(scorer, "n_fits", None)),
                             "structures": totals.get("structures", getattr(scorer, "n_structures", None)),
                             "seconds": time.time() - start_time}
        elapsed_time = np.round((time.time() - start_time) / 60, 1)
        print(f"Finished {len(self.out_names)} models (100%),", f"elapsed time: {elapsed_time} minutes")
        return None


def _default_batch(n: int) -> int:
    """Fits in flight per GPU: three N x N float64 matrices (+ scratch) per slot, at most ~96 GiB of the 288 GB
    of HBM and 512 slots (measured at N = 2048: 512 in flight is 2.5 % faster than 256, more adds nothing)."""
    npad = (n + 63) // 64 * 64
    per_slot = 3 * npad * npad * 8 + (npad // 2 + 64) ** 2 * 8 + npad * 64 * 8
    return int(max(2, min(512, (96 << 30) // per_slot)))
