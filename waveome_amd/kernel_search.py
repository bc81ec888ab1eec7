"""Compositional kernel search on top of the batched GPU fit.

Reference: waveome/model_search.py - `full_kernel_search` (:2987-3273), `loc_kernel_search`
(:2347-2558), `prod_kernel_creation` (:2561-2664), `check_if_better_metric` (:2667-2681),
`keep_top_k` (:2684-2710), `prune_best_model2` (:2778-2885), `prune_prod_kernel` (:2888-2984),
`kernel_test` (:2239-2334); waveome/utilities.py `check_if_model_exists` (:281-307).

Design.  In the reference every output column walks the search tree in its own Ray task and
calls `kernel_test` (build a model, fit it, score it) once per candidate, one after the other.
Every candidate starts from the default parameter values (the base kernel's trainable
parameters are re-assigned 1.0 at :2420-2421 and :2857-2858, new kernels are fresh copies, the
mean / likelihood are fresh), so a fit depends only on (structure, output column): the
candidates of one depth level are independent of each other.  Here the per-output search is a
*generator* that yields the list of candidates it wants scored wherever the reference loops
over `kernel_test`, and `run_searches` advances the generators of all output columns in lock
step, pooling their requests: one `Problem` per distinct structure, fitted for all the columns
that asked for it in one batched call (sharded over ranks when torch.distributed is up).

Deliberate deviations, all in code the reference itself flags as unfinished:
 * `prod_kernel_creation` re-inserts the *name* of a re-ordered product term at its sorted
   position but leaves the kernel where it was, so names and `kernel.kernels` disagree later in
   `prune_best_model2` (its TODO at :2816).  Here the kernel moves with its name.
 * `prune_prod_kernel` builds `np.array(other_kernel + [piece])[argsort([other_name, piece])]`,
   which keeps only the first two kernels when `other_kernel` has more than one (:2939-2944).
   Here all the other components are kept.
 * a depth level that yields no new candidate ends the search (the reference raises on `min([])`).
 * candidates whose fit fails (non-finite objective) are dropped, as the reference's
   `try / except: None` around `kernel_test` does.
"""
from __future__ import annotations

import copy
import json
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import kernels as K
from . import parallel
from .spec import ModelSpec
from .utilities import calc_bic, restart_points


# ---------------------------------------------------------------------------------------------
# names and candidate generation (host only)
# ---------------------------------------------------------------------------------------------
def kernel_info(k) -> str:
    """`k.name + str(k.active_dims)` (model_search.py:2402), "constant" for the static kernel."""
    if k.name == "constant":
        return k.name
    return k.name + "[" + " ".join(str(int(d)) for d in k.active_dims) + "]"


def set_feature_kernels(f: int, kern_list: Sequence, cat_vars: Sequence[int]) -> List:
    """model_search.py:2337-2344."""
    if f in cat_vars:
        return [K.Categorical(active_dims=[f])]
    k_list = list(kern_list)
    for k_ in k_list:
        k_.active_dims = [int(f)]
    return k_list


def _reset_trainable(kernel):
    for p in kernel.trainable_parameters:
        p.assign(1.0)


def _freeze_variance(k):
    try:
        K.set_trainable(k.variance, False)
    except AttributeError:
        K.set_trainable(k.base_kernel.variance, False)


class _Form:
    """A model structure the way the search manipulates it: an ordered list of additive terms, each an ordered list of
    (label, kernel) factors, label = `kernel_info` of the factor.  Candidate names and kernels are both RENDERED from
    this one form (`name`, `kernel()`), so they cannot drift apart.  The rules of one search step
    (model_search.py:2347-2664) are the three methods below; where a term or factor is placed is part of the
    reference's naming contract (names are compared as whole strings, terms are only ever prepended, appended or moved
    once - not sorted)."""

    def __init__(self, terms):
        self.terms = [list(t) for t in terms]

    @classmethod
    def parse(cls, kernel, name):
        comps = list(kernel.kernels) if kernel.name == "sum" else [kernel]
        terms = []
        for part, comp in zip(name.split("+"), comps):
            factors = list(comp.kernels) if comp.name == "product" else [comp]
            terms.append(list(zip(part.split("*"), factors)))
        return cls(terms)

    @staticmethod
    def _term_name(term):
        return "*".join(label for label, _ in term)

    @property
    def name(self):
        return "+".join(self._term_name(t) for t in self.terms)

    def kernel(self):
        comps = [K.Product([k for _, k in t]) if len(t) > 1 else t[0][1] for t in self.terms]
        return K.Sum(comps) if len(comps) > 1 else comps[0]

    def mentions_categorical(self, column, term=None):
        tag = "categorical[" + str(column) + "]"
        terms = self.terms if term is None else [self.terms[term]]
        return any(tag in label for t in terms for label, _ in t)

    # -- rule 1: add a term.  The new term goes behind the whole structure if the structure's name sorts before the
    #    new label, else in front of it.
    #    (`ties_behind`: the pruning step's variant of the rule - a structure whose name EQUALS the label also stays in
    #    front (an argsort of the two names, model_search.py:2933-2934); an empty structure is just the new term.)
    def plus(self, label, kernel, ties_behind=False):
        new = [(label, kernel)]
        if not self.terms:
            return _Form([new])
        behind = self.name <= label if ties_behind else self.name < label
        return _Form(self.terms + [new] if behind else [new] + self.terms)

    def without(self, i):
        """The structure without its i-th additive term (possibly empty)."""
        return _Form([t for j, t in enumerate(self.terms) if j != i])

    # -- rule 2: multiply term i by a factor.  Two-way interactions only; inside the product the factor whose label
    #    sorts first leads.  If the existing factor leads the product stays where the term was; if the new factor
    #    leads the product moves (see below).
    def times(self, i, label, kernel):
        term = self.terms[i]
        if len(term) != 1:
            return None
        if self._term_name(term) < label:
            terms = list(self.terms)
            terms[i] = term + [(label, kernel)]
            return _Form(terms)
        # the new factor leads: the position is looked up among ALL current term names (the old one included), then
        # the old term is removed and the product inserted at that index of the shortened list; "no later name" means
        # the last index of the original list
        names = [self._term_name(t) for t in self.terms]
        pos = next((j for j, n in enumerate(names) if label < n), len(names) - 1)
        terms = [t for j, t in enumerate(self.terms) if j != i]
        terms.insert(pos, [(label, kernel)] + term)
        return _Form(terms)

    def copy(self, reset=True):
        """Deep copy; with `reset` every trainable parameter restarts from 1.0 (model_search.py:2420-2421)."""
        out = _Form([[(label, copy.deepcopy(k)) for label, k in t] for t in self.terms])
        if reset:
            for t in out.terms:
                for _, k in t:
                    _reset_trainable(k)
        return out


def _structure_key(name: str):
    """What makes two names the same model for `check_if_model_exists`: the set of additive components, each as its
    sorted characters (utilities.py:281-307)."""
    return frozenset("".join(sorted(part)) for part in name.split("+"))


class _Seen:
    """Names of the models fitted so far, looked up by structure."""

    def __init__(self, names: Iterable[str] = ()):
        self.keys = {_structure_key(n) for n in (names or ())}

    def __contains__(self, name):
        return _structure_key(name) in self.keys


def _proposals(n_features, kern_list, cat_vars, depth):
    """(column, label, fresh kernel) of every base kernel a step may bring in, feature by feature: the kernel list on a
    numeric column, Categorical on a categorical one, and once - with the first column at depth 1 - the frozen
    near-zero constant that stands for "no structure" (model_search.py:2337-2344, 2388-2400)."""
    for f in range(n_features):
        fresh = set_feature_kernels(f, [copy.deepcopy(x) for x in kern_list], cat_vars)
        if f == 0 and depth == 1:
            nothing = K.Constant(variance=1e-6)
            K.set_trainable(nothing.variance, False)
            fresh = fresh + [nothing]
        for k in fresh:
            yield f, kernel_info(k), k


def prod_kernel_candidates(base_kernel, base_name: str, new_kernel, prev_models: Iterable[str]) -> Dict[str, object]:
    """Multiply each additive component of `base_kernel` by `new_kernel` (model_search.py:2561-2664): rule 2 applied
    to every term that is not yet a product and does not already carry the Categorical kernel of the new column."""
    seen = prev_models if isinstance(prev_models, _Seen) else _Seen(prev_models)
    base = _Form.parse(base_kernel, base_name)
    label, column = kernel_info(new_kernel), new_kernel.active_dims[0]
    out = {}
    for i in range(len(base.terms)):
        if base.mentions_categorical(column, term=i):
            continue
        form = base.copy(reset=False).times(i, label, copy.deepcopy(new_kernel))
        if form is None or form.name in seen:
            continue
        out[form.name] = form.kernel()
    return out


def loc_kernel_candidates(n_features: int, kern_list: Sequence, base_kern=None, base_name: Optional[str] = None,
                          cat_vars: Sequence[int] = (), depth: int = 0, operation: str = "sum",
                          prev_models: Iterable[str] = ()) -> Dict[str, object]:
    """Candidates of one local search step: name -> kernel (model_search.py:2347-2558 without the fits).

    Every proposal (`_proposals`) is combined with the base structure by `operation`:
      sum            rule 1 (add a term);
      product        rule 2 on a single base kernel (a product is never multiplied again);
      split_product  rule 2 on every term of a sum.
    A column whose Categorical kernel is already in the base brings nothing new (sum / product); a factor that joins
    a product has its variance frozen at 1 (:2469-2472); structures fitted before are not proposed again."""
    if operation not in ("sum", "product", "split_product"):
        raise ValueError(f"unknown operation {operation!r}")
    out: Dict[str, object] = {}
    if base_kern is None:
        return {label: k for _, label, k in _proposals(n_features, kern_list, cat_vars, depth)}
    seen = _Seen(prev_models)
    base = _Form.parse(base_kern, base_name)
    for f, label, k in _proposals(n_features, kern_list, cat_vars, depth):
        if operation == "split_product":
            _freeze_variance(k)
            out.update(prod_kernel_candidates(base.copy().kernel(), base_name, k, seen))
            continue
        if base.mentions_categorical(f):
            continue
        if operation == "sum":
            form = base.copy().plus(label, k)
        else:
            if "*" in base_name:
                continue
            _freeze_variance(k)
            form = base.copy().times(0, label, k)
        if form.name not in seen:
            out[form.name] = form.kernel()
    return out


def prune_candidates(best_name: str, best_kernel, existing: Iterable[str]) -> Dict[str, object]:
    """Structures `prune_best_model2` / `prune_prod_kernel` would refit (model_search.py:2778-2984), on the same `_Form`
    the search steps use: every additive term is dropped once (`without`); a product term is instead replaced by each of
    its factors in turn, the factor placed against the remaining terms by the pruning step's own rule (`plus` with ties
    going behind).  A single base kernel has nothing to prune; a lone product is split into its factors.  Every candidate
    restarts from 1.0 (:2857-2858); structures fitted before are not proposed again."""
    seen = existing if isinstance(existing, _Seen) else _Seen(existing)
    best = _Form.parse(best_kernel, best_name)
    out: Dict[str, object] = {}
    if len(best.terms) == 1 and len(best.terms[0]) == 1:
        return out
    for i, term in enumerate(best.terms):
        rest = best.copy().without(i)
        if len(term) == 1:
            forms = [rest] if rest.terms else []
        else:
            forms = [rest.copy().plus(label, copy.deepcopy(k), ties_behind=True) for label, k in best.copy().terms[i]]
        for form in forms:
            if form.name not in seen:
                out[form.name] = form.kernel()
    return out


def check_if_better_metric(model_dict: dict, depth: int) -> bool:
    """model_search.py:2667-2681."""
    prev_vals = [x["bic"] for x in model_dict.values() if x["depth"] == depth - 1]
    new_vals = [x["bic"] for x in model_dict.values() if x["depth"] == depth]
    if len(prev_vals) > 0 and len(new_vals) > 0:
        return bool(min(new_vals) < min(prev_vals))
    return False


def keep_top_k(res_dict: dict, depth: int, metric_diff: float = 6, split: bool = False) -> dict:
    """Stop expanding candidates more than `metric_diff` above the level's best (model_search.py:2684-2710)."""
    t = np.log(metric_diff) if split else metric_diff
    best_bic = min(v["bic"] for v in res_dict.values() if v["depth"] == depth)
    for v in res_dict.values():
        if v["depth"] == depth and v["bic"] - best_bic > t:
            v["try_next"] = False
    return res_dict


# ---------------------------------------------------------------------------------------------
# the per-output search, as a generator of fit requests
# ---------------------------------------------------------------------------------------------
def _entry(kernel, fit, depth, parent):
    return {"kernel": kernel, "bic": fit["bic"], "depth": depth, "parent": parent, "try_next": True, "fit": fit}


def _best_at(search: dict, depth: int):
    return min((v["bic"], v["depth"], k) for k, v in search.items() if v["depth"] == depth)[2]


def full_kernel_search_steps(n_features: int, kern_list: Sequence, cat_vars: Sequence[int] = (), max_depth: int = 5,
                             keep_all: bool = False, metric_diff: float = 6, early_stopping: bool = True,
                             prune: bool = True, keep_only_best: bool = True):
    """Generator form of `full_kernel_search` (model_search.py:2987-3273).

    Yields dict name -> kernel (candidates to score) and expects `.send({name: fit})` back, where
    fit = {"bic": ..., ...} or None for a failed fit.  Returns (StopIteration.value) the dict the
    reference returns, without "var_exp" (computed by the caller from the fitted best model)."""
    search: dict = {}
    edges: list = []

    def do_prune(d):
        best_bic, best_name = min((v["bic"], k) for k, v in search.items() if v["depth"] == d)
        cands = prune_candidates(best_name, search[best_name]["kernel"], search.keys())
        return best_bic, best_name, cands

    d = 0
    for d in range(1, max_depth + 1):
        if d == 1:
            cands = loc_kernel_candidates(n_features, kern_list, cat_vars=cat_vars, depth=d)
            parents = {name: "None" for name in cands}
        else:
            cands, parents = {}, {}
            seen = list(search.keys())
            for k in list(search.keys()):
                if search[k]["depth"] != d - 1 or search[k]["try_next"] is False or k == "constant":
                    continue
                cur = search[k]["kernel"]
                for op in ("sum", "split_product" if cur.name == "sum" else "product"):
                    new = loc_kernel_candidates(n_features, kern_list, base_kern=cur, base_name=k,
                                                cat_vars=cat_vars, depth=d, operation=op, prev_models=seen)
                    for name, kern in new.items():
                        cands[name], parents[name] = kern, k
                        edges.append((k, name))
                    seen += list(new.keys())
        fits = (yield cands) if cands else {}
        for name, kern in cands.items():
            if fits.get(name) is not None:
                search[name] = _entry(kern, fits[name], d, parents[name])
        if not any(v["depth"] == d for v in search.values()):
            break
        best_name = _best_at(search, d)
        if best_name == "constant":
            break
        stop = early_stopping and d > 1 and not check_if_better_metric(search, d)
        if not stop and d != max_depth and not keep_all:
            keep_top_k(search, depth=d, metric_diff=metric_diff)
        if prune:
            best_bic, best_name, pc = do_prune(d)
            pfits = (yield pc) if pc else {}
            for name, kern in pc.items():
                f = pfits.get(name)
                if f is not None and f["bic"] < best_bic:
                    search[name] = _entry(kern, f, d, best_name)
        if stop:
            break
    best_model_name = min((v["bic"], v["depth"], k) for k, v in search.items())[2]
    models = {best_model_name: search[best_model_name]} if keep_only_best else search
    return {"models": models, "edges": edges, "best_model": best_model_name, "depth_reached": d}


# ---------------------------------------------------------------------------------------------
# pooled scoring of candidates on the GPU
# ---------------------------------------------------------------------------------------------
def structure_signature(spec: ModelSpec) -> str:
    return json.dumps(spec.to_dict(), sort_keys=True)


class GpuScorer:
    """Scores (column, kernel) candidates: all candidates of a step, whatever their structure, in one batched
    work queue on the GPU.  This is `kernel_test` (model_search.py:2239-2334) for penalization_factor = 0,
    `num_restart <= 1`: L-BFGS-B from the default values, `bic = round(2k - 2 log_posterior, 2)` with k =
    number of trainable Parameter objects (q_mu and q_sqrt of the reference's SVGP included, hence + 2)."""

    def __init__(self, X, Y, context=None, max_batch=None, num_opt_iter=50000, workers=None, mode=None,
                 num_restart=1, random_seed=None, shared=None):
        from .engine import lbfgs_opts
        from .model_classes import default_context

        # `shared`: the inputs as parallel.broadcast_inputs left them (device tensors of the RCCL broadcast under
        # nccl); otherwise the arrays are broadcast here so that every rank scores on rank 0's data
        # Missing values, as full_kernel_search treats them (model_search.py:3034-3043: per output, the rows where
        # every input and that output are observed; GPSearch itself refuses NaN, so only direct callers get here):
        # rows with a missing input go for every output; a missing output value masks the row for that output only,
        # on the device (wv_problem_caps::nan_is_missing), so all outputs keep sharing one batch.
        if shared is None:
            X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
            x_ok = ~np.isnan(X).any(axis=1)
            shared = parallel.broadcast_inputs(X[x_ok], Y[x_ok])
        self.shared = shared
        self.X, self.Y = self.shared.X, self.shared.Y
        self.nan_is_missing = bool(np.isnan(self.Y).any())
        self.ctx = context or default_context()
        self.max_batch = max_batch
        self.num_restart, self.random_seed = max(1, int(num_restart)), random_seed
        # A search step is many structures with a handful of columns each.  mode "multi" (default): one
        # multi-structure problem per step (wv_problem_create_multi / wv_fit_multi) - every launch mixes
        # structures, the batch stays full.  mode "threads": one Problem per structure, `workers` of them in
        # flight from host threads (kept for comparison: 3x faster than one at a time, 2.5x slower than "multi").
        self.mode = mode or os.environ.get("WV_SEARCH_MODE", "multi")
        if self.mode not in ("multi", "threads"):
            raise ValueError("mode must be 'multi' or 'threads'")
        self.workers = int(workers if workers is not None else os.environ.get("WV_SEARCH_WORKERS", 16))
        self.opts = lbfgs_opts(maxiter=num_opt_iter)
        self.n_fits = 0
        self.n_structures = 0
        self.cache: Dict[tuple, dict] = {}

    def _spec(self, kernel) -> ModelSpec:
        return ModelSpec(kernel, self.X.shape[1], K.GaussianLikelihood(), K.ConstantMean())

    def __call__(self, tasks: List[tuple], local: bool = False) -> List[Optional[dict]]:
        """tasks: [(column, kernel)] -> [fit dict or None] in the same order.

        `local`: score on THIS rank only, no `parallel.*` call - for callers that are not in lock step with the other
        ranks (the unsynchronised sharded search's fallback for a structure beyond its session's capacities: only the
        one rank gets there, so a collective would be unmatched and the job would hang or pair with another rank's
        later gather)."""
        from .engine import Problem
        from .model_search import _default_batch

        specs = [self._spec(kern) for _, kern in tasks]
        sigs = [structure_signature(sp) for sp in specs]
        out: List[Optional[dict]] = [None] * len(tasks)
        todo = [i for i in range(len(tasks)) if (tasks[i][0], sigs[i]) not in self.cache]
        # group by structure; keep first-seen order so every rank builds the same task list
        groups: Dict[str, List[int]] = {}
        for i in todo:
            groups.setdefault(sigs[i], []).append(i)
        order = [i for g in groups.values() for i in g]
        width = max([specs[i].n_trainable for i in order], default=0)
        costs = np.array([specs[i].n_trainable for i in order], dtype=np.float64)
        N = self.X.shape[0]

        def fit_shard(ids):
            rec = np.full((len(ids), 5 + width), np.nan)
            mine = [order[j] for j in ids]
            by_sig: Dict[str, List[int]] = {}
            for row, i in enumerate(mine):
                by_sig.setdefault(sigs[i], []).append(row)
            if self.mode == "multi":
                # ONE multi-structure problem and one work queue for the whole shard: every launch mixes
                # structures, so the batch stays full however few columns ask for any one structure
                sig_list = list(by_sig)
                sp_list = [specs[mine[by_sig[sg][0]]] for sg in sig_list]
                sidx = np.empty(len(mine), dtype=np.int32)
                for k, sg in enumerate(sig_list):
                    sidx[by_sig[sg]] = k
                cols = np.array([tasks[i][0] for i in mine], dtype=np.int32)
                R = self.num_restart  # R > 1: seeded random starts, the best LML is kept (kernel_test :2283-2288)
                rep = np.repeat(np.arange(len(mine)), R)
                cols_r, sidx_r = cols[rep], sidx[rep]
                U0 = np.zeros((len(rep), width))
                for k, sp in enumerate(sp_list):
                    rows = np.nonzero(sidx_r == k)[0]
                    if R == 1:
                        U0[rows, :sp.n_trainable] = sp.unconstrain()
                    else:
                        pts = restart_points(sp.n_trainable, R, random_seed=self.random_seed)
                        U0[rows, :sp.n_trainable] = np.tile(pts, (len(rows) // R, 1))
                mb = self.max_batch or min(len(rep), _default_batch(N))
                prob = self.shared.problem(self.ctx, sp_list, max(1, mb), nan_is_missing=self.nan_is_missing)
                try:
                    res = prob.fit_multi(cols_r, sidx_r, U0, self.opts)
                finally:
                    prob.close()
                lml = np.where(np.isfinite(res["lml"]), res["lml"], -np.inf).reshape(len(mine), R)
                best = np.arange(len(mine)) * R + np.argmax(lml, axis=1)
                rec[:, :5] = np.column_stack([res["loss"][best], res["lml"][best], res["nit"][best],
                                              res["nfev"][best], res["status"][best]])
                rec[:, 5:5 + width] = res["u"][best, :width]
                return rec

            if self.num_restart > 1:
                raise NotImplementedError("random restarts are implemented in mode 'multi' only")

            def fit_structure(rows):
                sp = specs[mine[rows[0]]]
                cols = np.array([tasks[mine[r]][0] for r in rows], dtype=np.int32)
                mb = self.max_batch or min(len(rows), _default_batch(N))
                prob = self.shared.problem(self.ctx, sp, max(1, mb), nan_is_missing=self.nan_is_missing)
                try:
                    res = prob.fit(cols, np.tile(sp.unconstrain(), (len(rows), 1)), self.opts)
                finally:
                    prob.close()
                rec[rows, :5] = np.column_stack([res["loss"], res["lml"], res["nit"], res["nfev"], res["status"]])
                rec[rows, 5:5 + sp.n_trainable] = res["u"]

            # mode "threads": one Problem per structure, several in flight from host threads (each has its own
            # HIP streams; the C-ABI calls release the GIL).  Heaviest first; memory bounded by workers x Problem
            work = sorted(by_sig.values(), key=lambda rows: -len(rows) * specs[mine[rows[0]]].n_trainable)
            per_problem = max(1, min(max(len(r) for r in work), _default_batch(N))) * 2 * (N + 64) ** 2 * 8
            nw = max(1, min(self.workers, len(work), (48 << 30) // per_problem))
            if nw <= 1:
                for rows in work:
                    fit_structure(rows)
            else:
                with ThreadPoolExecutor(max_workers=nw) as pool:
                    for f in [pool.submit(fit_structure, rows) for rows in work]:
                        f.result()
            return rec

        def fit_stream(dealer):
            """This rank's share of a step dealt in chunks (parallel.dynamic_map; candidate fits run to SciPy's
            tolerances and differ several-fold in cost): one problem holding every structure of the step, one work
            queue kept full across the chunks drawn.  A task is one candidate: its R restarts travel together."""
            sig_list = list(groups)
            sig_pos = {sg: k for k, sg in enumerate(sig_list)}
            sp_list = [specs[groups[sg][0]] for sg in sig_list]
            R = self.num_restart
            mb = self.max_batch or min(-(-len(order) * R // world) + dyn_chunk * R, _default_batch(N))
            prob = self.shared.problem(self.ctx, sp_list, max(1, mb), nan_is_missing=self.nan_is_missing)

            def pull_fits():  # candidate ids -> fit ids (candidate * R + restart)
                c = np.asarray(dealer.pull(), dtype=np.int64)
                return (c[:, None] * R + np.arange(R)[None, :]).reshape(-1)

            def task_args(fit_ids):
                cand = [order[int(f) // R] for f in fit_ids]
                sidx = np.array([sig_pos[sigs[i]] for i in cand], dtype=np.int32)
                U0 = np.zeros((len(fit_ids), width))
                for row, (f, i) in enumerate(zip(fit_ids, cand)):
                    sp = specs[i]
                    U0[row, :sp.n_trainable] = (sp.unconstrain() if R == 1 else
                                                restart_points(sp.n_trainable, R, random_seed=self.random_seed)[int(f) % R])
                return np.array([tasks[i][0] for i in cand], dtype=np.int32), sidx, U0, None

            try:
                fit_ids, res = prob.fit_stream(pull_fits, task_args, self.opts, width=width, backlog=mb // 4,
                                               first_fill=dealer.first_fill * R)
            finally:
                prob.close()
            cand_ids = fit_ids.reshape(-1, R)[:, 0] // R
            lml = np.where(np.isfinite(res["lml"]), res["lml"], -np.inf).reshape(-1, R)
            best = np.arange(len(cand_ids)) * R + np.argmax(lml, axis=1)
            rec = np.full((len(cand_ids), 5 + width), np.nan)
            rec[:, :5] = np.column_stack([res["loss"][best], res["lml"][best], res["nit"][best], res["nfev"][best],
                                          res["status"][best]])
            rec[:, 5:5 + width] = res["u"][best, :width]
            return cand_ids, rec

        if order:
            rank, world = parallel.dist_info()
            dyn_chunk = max(1, min(_default_batch(N) // 16, -(-len(order) // (8 * world))))
            if local:
                rec = fit_shard(np.arange(len(order)))
            elif parallel.dist_active() and self.mode == "multi":
                rec = parallel.dynamic_map(len(order), fit_stream, dyn_chunk, 5 + width, costs=costs)
                self.rank_stats = parallel.last_map_stats()
            elif parallel.dist_active():
                rec = parallel.sharded_map(len(order), fit_shard, costs=costs, width=5 + width)
            else:
                rec = fit_shard(np.arange(len(order)))
            self.n_fits += len(order)
            self.n_structures += len(groups)
            for j, i in enumerate(order):
                sp = specs[i]
                loss = rec[j, 0]
                if not np.isfinite(loss):
                    self.cache[(tasks[i][0], sigs[i])] = None
                    continue
                k_count = sum(1 for q in sp.param_objs if q.trainable) + 2
                self.cache[(tasks[i][0], sigs[i])] = {
                    "bic": round(calc_bic(loglik=-loss, n=N, k=k_count), 2), "loss": float(loss),
                    "lml": float(rec[j, 1]), "nit": int(rec[j, 2]), "nfev": int(rec[j, 3]),
                    "status": int(rec[j, 4]), "u": rec[j, 5:5 + sp.n_trainable].copy()}
        for i in range(len(tasks)):
            out[i] = self.cache[(tasks[i][0], sigs[i])]
        return out


class _SearchSession:
    """One growable multi-structure problem and its work queue for a whole search (wv_problem_create_ex /
    wv_problem_add_spec / wv_queue_*): candidates are submitted as they are asked for and scored fits come back as
    they finish, so an output column never waits for another column's depth level."""

    # room for the structures of a search: every structure must fit these (one that does not is scored through the
    # blocking path).  Kept moderate on purpose - the Gram / trace kernels size their LDS by max_feat / max_red.
    CAPS = {"max_specs": 4096, "max_params": 40, "max_feat": 24, "max_factors": 24, "max_red": 48}

    def __init__(self, scorer: "GpuScorer"):
        from .engine import FitQueue
        from .model_search import _default_batch

        self.sc = scorer
        self.N = scorer.X.shape[0]
        self.sig_index: Dict[str, int] = {}
        self.waiters: Dict[tuple, list] = {}   # (col, sig) -> [callback(fit)] while the fits are in flight
        self.tasks: Dict[int, tuple] = {}      # tag -> (col, sig, restart index)
        self.partial: Dict[tuple, list] = {}   # (col, sig) -> results of its restarts so far
        self.next_tag = 0
        self.prob = None
        self.queue = None
        self._FitQueue = FitQueue
        self.mb = scorer.max_batch or _default_batch(self.N)

    def _open(self, first_spec: ModelSpec):
        self.prob = self.sc.shared.problem(self.sc.ctx, [first_spec], max(1, self.mb), capacity=self.CAPS,
                                           nan_is_missing=self.sc.nan_is_missing)
        self.queue = self._FitQueue(self.prob, self.sc.opts, width=self.CAPS["max_params"])

    def request(self, col: int, kernel, callback):
        """Score (col, kernel); `callback(fit or None)` fires at once on a cache hit, else when the fit is done."""
        from . import _lib

        sc = self.sc
        spec = sc._spec(kernel)
        sig = structure_signature(spec)
        key = (col, sig)
        if key in sc.cache:
            callback(sc.cache[key])
            return
        if key in self.waiters:
            self.waiters[key].append(callback)
            return
        if sig not in self.sig_index:
            try:
                if self.prob is None:
                    self._open(spec)
                    self.sig_index[sig] = 0
                else:
                    self.sig_index[sig] = self.prob.add_spec(spec)
                sc.n_structures += 1
            except _lib.WaveomeAmdError:
                # beyond the session's capacities: the blocking path, in its own problem - on this rank alone (a
                # session belongs to one rank; under torch.distributed the other ranks are elsewhere in their searches)
                callback(sc([(col, kernel)], local=True)[0])
                return
        sidx = self.sig_index[sig]
        R = sc.num_restart
        self.waiters[key] = [callback]
        self.partial[key] = []
        U0 = np.zeros((R, self.queue.width))
        nt = spec.n_trainable
        U0[:, :nt] = spec.unconstrain() if R == 1 else restart_points(nt, R, random_seed=sc.random_seed)
        tags = np.arange(self.next_tag, self.next_tag + R, dtype=np.int64)
        for r, tg in enumerate(tags):
            self.tasks[int(tg)] = (col, sig, spec)
        self.next_tag += R
        self.queue.submit(np.full(R, col, dtype=np.int32), np.full(R, sidx, dtype=np.int32), U0, tags)
        sc.n_fits += 1

    def busy(self) -> bool:
        return self.queue is not None and (self.queue.in_system > 0 or self.queue.done > 0)

    def pump(self):
        """One scheduling step of the queue; fires the callbacks of the candidates that finished."""
        if self.queue is None:
            return
        done, _ = self.queue.advance()
        if not done:
            return
        res = self.queue.fetch(max(done, 64))
        sc = self.sc
        for j, tg in enumerate(res["tag"]):
            col, sig, spec = self.tasks.pop(int(tg))
            key = (col, sig)
            self.partial[key].append((res["lml"][j], res["loss"][j], res["nit"][j], res["nfev"][j], res["status"][j],
                                      res["u"][j, :spec.n_trainable].copy()))
            if len(self.partial[key]) < sc.num_restart:
                continue
            runs = self.partial.pop(key)
            # restarts: the fit with the best maximum_log_likelihood_objective is scored (kernel_test :2283-2288)
            best = max(runs, key=lambda r: r[0] if np.isfinite(r[0]) else -np.inf)
            lml, loss, nit, nfev, status, u = best
            if not np.isfinite(loss):
                fit = None
            else:
                k_count = sum(1 for q in spec.param_objs if q.trainable) + 2
                fit = {"bic": round(calc_bic(loglik=-loss, n=self.N, k=k_count), 2), "loss": float(loss),
                       "lml": float(lml), "nit": int(nit), "nfev": int(nfev), "status": int(status), "u": u}
            sc.cache[key] = fit
            for cb in self.waiters.pop(key):
                cb(fit)

    def close(self):
        if self.queue is not None:
            self.queue.close()
        if self.prob is not None:
            self.prob.close()
        self.queue = self.prob = None


def run_searches_async(columns: Sequence[int], n_features: int, kern_list: Sequence, scorer: "GpuScorer",
                       cat_vars: Sequence[int] = (), verbose: bool = False, session=None,
                       more_columns: Optional[Callable[[], Sequence[int]]] = None, keep_active: int = 0,
                       **search_kwargs) -> Dict[int, dict]:
    """The searches of `run_searches` without its lock step: every output column's generator gets the fits it asked
    for as soon as they are all done and its next candidates join the running work queue at once - as in the
    reference, where each column is its own Ray task (model_search.py:1200-1222).  Same fits (a fit depends only on
    (structure, column)), hence the same search trees and best models.
    `more_columns` / `keep_active`: whenever fewer than `keep_active` searches are running, more columns are drawn
    from `more_columns()` (empty: none left) - how the ranks of a job share the columns (`run_searches_sharded`).
    `session`: anything with request(col, kernel, callback) / busy() / pump() / close() (test seam)."""
    session = session or _SearchSession(scorer)
    gens, results = {}, {}
    pending: Dict[int, dict] = {}  # col -> {"names": [...], "fits": {name: fit}, "left": int}
    ready: List[int] = []

    def ask(c, cands):
        st = {"names": list(cands), "fits": {}, "left": len(cands)}
        pending[c] = st

        def make_cb(name):
            def cb(fit):
                st["fits"][name] = fit
                st["left"] -= 1
                if st["left"] == 0:
                    ready.append(c)
            return cb

        for name, kern in cands.items():
            session.request(c, kern, make_cb(name))
        if not cands:
            ready.append(c)

    def start(c):
        g = full_kernel_search_steps(n_features, kern_list, cat_vars=cat_vars, **search_kwargs)
        try:
            req = next(g)
            gens[c] = g
            ask(c, req)
        except StopIteration as e:  # pragma: no cover - a search always scores depth 1
            results[c] = e.value

    source = [more_columns]

    def top_up():
        while source[0] is not None and len(pending) < keep_active:
            new = [int(c) for c in source[0]()]
            if not new:
                source[0] = None
                break
            for c in new:
                start(c)

    try:
        for c in columns:
            start(int(c))
        top_up()
        steps = 0
        while pending:
            while ready:
                c = ready.pop()
                st = pending.pop(c)
                try:
                    ask(c, gens[c].send(st["fits"]))
                except StopIteration as e:
                    results[c] = e.value
            top_up()
            if ready:
                continue
            if not pending:
                break
            if not session.busy():
                raise RuntimeError("search stalled: candidates pending but no fit in flight")  # pragma: no cover
            session.pump()
            steps += 1
        if verbose:
            print(f"asynchronous search: {getattr(scorer, 'n_fits', '?')} candidate fits over "
                  f"{getattr(scorer, 'n_structures', '?')} structures in {steps} queue steps")
    finally:
        session.close()
    return results


def run_searches_sharded(n_columns: int, n_features: int, kern_list: Sequence, scorer: "GpuScorer",
                         cat_vars: Sequence[int] = (), verbose: bool = False, session=None,
                         **search_kwargs) -> Dict[int, dict]:
    """The unsynchronised search under torch.distributed: every output column's search runs entirely on ONE rank (its
    generator never crosses ranks, so no level barrier anywhere - the reference's one Ray task per column,
    model_search.py:1200-1222), the columns are dealt to the ranks on demand (`parallel.TaskDealer`: a rank starts
    with three quarters of its fair share and draws more as its searches finish), and the per-column results are
    all-gathered at the end (`parallel.gather_objects`).  Same fits as everywhere else, hence the same trees.
    Per-rank busy times -> `parallel.last_map_stats()`."""
    import time

    rank, world = parallel.dist_info()
    fair = -(-n_columns // world)
    keep_active = max(1, -(-3 * fair // 4))
    dealer = parallel.TaskDealer(n_columns, max(1, keep_active // 4))
    t0 = time.perf_counter()
    mine = run_searches_async([], n_features, kern_list, scorer, cat_vars=cat_vars, verbose=verbose, session=session,
                              more_columns=dealer.pull, keep_active=keep_active, **search_kwargs)
    busy = time.perf_counter() - t0
    merged: Dict[int, dict] = {}
    totals = [0, 0]
    for part, n_fits, n_struct in parallel.gather_objects((mine, getattr(scorer, "n_fits", 0),
                                                           getattr(scorer, "n_structures", 0))):
        merged.update(part)
        totals[0] += int(n_fits or 0)
        totals[1] += int(n_struct or 0)
    dealer.close()
    parallel._record_stats("dynamic-columns", busy, len(mine))
    scorer.job_totals = {"fits": totals[0], "structures": totals[1]}
    if sorted(merged) != list(range(n_columns)):
        raise RuntimeError("sharded search: the ranks' column sets do not add up to the job")
    return merged


def run_searches(columns: Sequence[int], n_features: int, kern_list: Sequence, scorer: Callable,
                 cat_vars: Sequence[int] = (), verbose: bool = False, **search_kwargs) -> Dict[int, dict]:
    """Advance one `full_kernel_search_steps` generator per output column in lock step, pooling the
    candidates of all columns into one `scorer` call per step."""
    gens, requests, results = {}, {}, {}
    for c in columns:
        g = full_kernel_search_steps(n_features, kern_list, cat_vars=cat_vars, **search_kwargs)
        try:
            requests[c] = next(g)
            gens[c] = g
        except StopIteration as e:  # pragma: no cover - a search always scores depth 1
            results[c] = e.value
    step = 0
    while requests:
        tasks, owner = [], []
        for c, cands in requests.items():
            for name, kern in cands.items():
                tasks.append((c, kern))
                owner.append((c, name))
        fits = scorer(tasks)
        step += 1
        if verbose:
            print(f"search step {step}: {len(tasks)} candidate fits over {len(requests)} outputs")
        per_col: Dict[int, dict] = {c: {} for c in requests}
        for (c, name), f in zip(owner, fits):
            per_col[c][name] = f
        nxt = {}
        for c in list(requests):
            try:
                nxt[c] = gens[c].send(per_col[c])
            except StopIteration as e:
                results[c] = e.value
        requests = nxt
    return results
