"""Thin Python handles over the C-ABI device path (include/waveome_amd.h).

`Context` = one GPU; `Problem` = (X, Y, kernel spec) resident in HBM with workspace for
`max_batch` fits in flight.  All heavy work happens in libwaveome_amd.so; these classes only
marshal NumPy arrays.  There is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .spec import ModelSpec


def lbfgs_opts(maxiter=50000, maxfun=None, ftol=2.220446049250313e-09, gtol=1e-05, m=10, maxls=20):
    """SciPy L-BFGS-B option set; defaults = the reference's (waveome/model_classes.py:310-313)."""
    o = _lib.LbfgsOpts()
    o.m, o.maxiter, o.maxfun, o.maxls = int(m), int(maxiter), int(maxfun or maxiter), int(maxls)
    o.ftol, o.gtol = float(ftol), float(gtol)
    return o


class Context:
    """One per GPU and precision.  dtype "float64" (default) or "float32": the storage / MFMA type of the
    N x N work matrices - the reference's `set_precision` (waveome/utilities.py:34-49); parameters, vectors,
    reductions and every host-side number stay float64."""

    def __init__(self, device: int = 0, dtype: str = "float64"):
        if dtype not in ("float64", "float32"):
            raise ValueError("Precision must be either 'float32' or 'float64'")
        self.lib = _lib.load()
        self.handle = C.c_void_p()
        _lib.check(self.lib.wv_ctx_create(int(device), 1 if dtype == "float32" else 0, C.byref(self.handle)),
                   "wv_ctx_create")
        self.device = int(device)
        self.dtype = dtype

    def close(self):
        if self.handle:
            self.lib.wv_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _lib.check(self.lib.wv_ctx_synchronize(self.handle), "wv_ctx_synchronize")

    # ---- rate probes (include/waveome_amd_probes.h): only in the `make probes` build, not in the release library ----
    def _probe(self, name):
        fn = getattr(self.lib, name, None)
        if fn is None:
            raise _lib.WaveomeAmdError(
                f"{name} is a probe of the diagnostic build: `make -C waveome_amd/csrc probes`, then run with "
                "WAVEOME_AMD_LIB=tools/libwv_probes.so (tools/_probes_lib.py does the latter)")
        return fn

    def mfma_f64_rate(self, iters=20000, waves_per_simd=1):
        tf, ms = C.c_double(), C.c_double()
        _lib.check(self._probe("wv_dbg_mfma_f64_rate")(self.handle, iters, waves_per_simd, C.byref(tf),
                                                 C.byref(ms)), "wv_dbg_mfma_f64_rate")
        return tf.value, ms.value

    def valu_f64_rate(self, iters=20000, waves_per_simd=2, mode=0):
        v, m = C.c_double(), C.c_double()
        _lib.check(self._probe("wv_dbg_valu_f64_rate")(self.handle, iters, waves_per_simd, mode, C.byref(v),
                                                 C.byref(m)), "wv_dbg_valu_f64_rate")
        return v.value, m.value

    def gemm_bench(self, M, N, K, a_kcontig=True, b_kcontig=True, bn=128, nbatch=1, iters=5):
        tf, ms = C.c_double(), C.c_double()
        _lib.check(self._probe("wv_dbg_gemm_bench")(self.handle, M, N, K, int(a_kcontig), int(b_kcontig), bn,
                                              nbatch, iters, C.byref(tf), C.byref(ms)), "wv_dbg_gemm_bench")
        return tf.value, ms.value

    def copy_rate(self, nbytes=1 << 30, iters=10):
        g = C.c_double()
        _lib.check(self._probe("wv_dbg_copy_rate")(self.handle, int(nbytes), iters, C.byref(g)), "wv_dbg_copy_rate")
        return g.value

    # ---- test hooks of the release library (include/waveome_amd_debug.h) ----
    def dbg_gemm(self, A, B, Cm, a_kcontig=True, b_kcontig=True, bn=128, alpha=1.0, beta=0.0):
        """A: [nb, M, K] (or [nb, K, M] if not a_kcontig); B likewise; Cm: [nb, M, N] updated copy."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        out = np.array(Cm, dtype=np.float64, order="C", copy=True)
        nb, M, N = out.shape
        K = A.shape[2] if a_kcontig else A.shape[1]
        _lib.check(self.lib.wv_dbg_gemm(self.handle, _lib.as_dp(A), _lib.as_dp(B), _lib.as_dp(out), M, N, K,
                                        int(a_kcontig), int(b_kcontig), bn, alpha, beta, nb), "wv_dbg_gemm")
        return out


class Problem:
    def __init__(self, ctx: Context, X, Y, spec: ModelSpec, max_batch: int = 8,
                 x_device_ptr: Optional[int] = None, y_device_ptr: Optional[int] = None,
                 shape: Optional[Sequence[int]] = None, capacity: Optional[dict] = None,
                 nan_is_missing: bool = False):
        """X: (N, D) array, Y: (N, P) array (one column per output).  Alternatively pass raw device
        pointers (x_device_ptr -> N x D row-major, y_device_ptr -> P x N row-major) with
        shape=(N, D, P), e.g. buffers filled by an RCCL broadcast.
        nan_is_missing: a NaN in Y[:, p] marks that row as missing for output p - fits, LML and gradients of that
        output are those of its complete cases (the reference's per-output filter, model_search.py:3034-3043) while
        all outputs keep sharing one batch; gram / predict / component means are then refused."""
        self.ctx, self.lib = ctx, ctx.lib
        # a list of specs makes a multi-structure problem (fit_multi only): tasks of different kernel
        # structures share the workspace and one work queue
        self.specs = list(spec) if isinstance(spec, (list, tuple)) else [spec]
        self.spec = spec = self.specs[0]
        self._cspecs = [sp.to_ctypes() for sp in self.specs]
        self._cspec = self._cspecs[0]
        self.handle = C.c_void_p()
        if x_device_ptr is not None:
            N, D, P = (int(v) for v in shape)
            xp, yp, xd, yd = C.c_void_p(x_device_ptr), C.c_void_p(y_device_ptr), 1, 1
        else:
            X = np.ascontiguousarray(X, dtype=np.float64)
            Y = np.asarray(Y, dtype=np.float64)
            if Y.ndim == 1:
                Y = Y[:, None]
            N, D = X.shape
            if Y.shape[0] != N:
                raise ValueError("X and Y disagree on N")
            P = Y.shape[1]
            self._Yt = np.ascontiguousarray(Y.T)
            self._X = X
            xp, yp, xd, yd = C.c_void_p(X.ctypes.data), C.c_void_p(self._Yt.ctypes.data), 0, 0
        self.N, self.D, self.P, self.max_batch = N, D, P, int(max_batch)
        if capacity is not None or nan_is_missing:
            # room for structures added later (add_spec): max_specs / max_params / max_feat / max_factors / max_red
            caps = _lib.ProblemCaps(**{k: int((capacity or {}).get(k, 0)) for k in
                                       ("max_specs", "max_params", "max_feat", "max_factors", "max_red")})
            caps.nan_is_missing = 1 if nan_is_missing else 0
            arr = (C.POINTER(_lib.KernelSpec) * len(self.specs))(*[C.pointer(cs) for cs in self._cspecs])
            _lib.check(self.lib.wv_problem_create_ex(ctx.handle, xp, N, D, yp, P, arr, len(self.specs), C.byref(caps),
                                                     int(max_batch), xd, yd, C.byref(self.handle)),
                       "wv_problem_create_ex")
        elif len(self.specs) == 1:
            _lib.check(self.lib.wv_problem_create(ctx.handle, xp, N, D, yp, P, C.byref(self._cspec),
                                                  int(max_batch), xd, yd, C.byref(self.handle)),
                       "wv_problem_create")
        else:
            arr = (C.POINTER(_lib.KernelSpec) * len(self.specs))(*[C.pointer(cs) for cs in self._cspecs])
            _lib.check(self.lib.wv_problem_create_multi(ctx.handle, xp, N, D, yp, P, arr, len(self.specs),
                                                        int(max_batch), xd, yd, C.byref(self.handle)),
                       "wv_problem_create_multi")
        self.n_params = spec.n_params
        self.n_trainable = spec.n_trainable

    def close(self):
        if self.handle:
            self.lib.wv_problem_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_spec(self, spec: ModelSpec) -> int:
        """Add a kernel structure to a problem created with `capacity`; returns its spec index."""
        cs = spec.to_ctypes()
        idx = C.c_int32()
        _lib.check(self.lib.wv_problem_add_spec(self.handle, C.byref(cs), C.byref(idx)), "wv_problem_add_spec")
        self.specs.append(spec)
        self._cspecs.append(cs)
        return int(idx.value)

    @property
    def workspace_bytes(self):
        return int(self.lib.wv_problem_workspace_bytes(self.handle))

    def gram(self, theta) -> np.ndarray:
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        K = np.empty((self.N, self.N))
        _lib.check(self.lib.wv_gram(self.handle, _lib.as_dp(theta), _lib.as_dp(K)), "wv_gram")
        return K

    def lml(self, cols, thetas):
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        B = len(cols)
        out, st = np.empty(B), np.empty(B, dtype=np.int32)
        _lib.check(self.lib.wv_lml(self.handle, _lib.as_ip(cols), B, _lib.as_dp(thetas), _lib.as_dp(out),
                                   _lib.as_ip(st)), "wv_lml")
        return out, st

    def value_grad(self, cols, U):
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        U = np.ascontiguousarray(np.atleast_2d(U), dtype=np.float64)
        B = len(cols)
        loss, grad, st = np.empty(B), np.empty((B, self.n_trainable)), np.empty(B, dtype=np.int32)
        _lib.check(self.lib.wv_value_grad(self.handle, _lib.as_ip(cols), B, _lib.as_dp(U), _lib.as_dp(loss),
                                          _lib.as_dp(grad), _lib.as_ip(st)), "wv_value_grad")
        return loss, grad, st

    def fit(self, cols, U0, opts=None, prior_scale=None):
        """Fit len(cols) tasks; returns dict of arrays (u, loss, lml, nit, nfev, status).
        prior_scale: optional per-task Horseshoe scale (1 / penalization_factor of that output)."""
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        U = np.array(np.atleast_2d(U0), dtype=np.float64, order="C", copy=True)
        T = len(cols)
        if U.shape != (T, self.n_trainable):
            raise ValueError(f"U0 must be ({T}, {self.n_trainable})")
        opts = opts or lbfgs_opts()
        loss, lml = np.empty(T), np.empty(T)
        nit, nfev, st = (np.empty(T, dtype=np.int32) for _ in range(3))
        ps = None
        if prior_scale is not None:
            ps = np.ascontiguousarray(np.broadcast_to(prior_scale, (T,)), dtype=np.float64)
        _lib.check(self.lib.wv_fit_penalized(self.handle, _lib.as_ip(cols), T, C.byref(opts),
                                             _lib.as_dp(ps) if ps is not None else None, _lib.as_dp(U),
                                             _lib.as_dp(loss), _lib.as_dp(lml), _lib.as_ip(nit),
                                             _lib.as_ip(nfev), _lib.as_ip(st)), "wv_fit_penalized")
        return {"u": U, "loss": loss, "lml": lml, "nit": nit, "nfev": nfev, "status": st}

    def fit_multi(self, cols, spec_idx, U0, opts=None, prior_scale=None):
        """Fit tasks that differ in kernel structure as well as in column: task t uses self.specs[spec_idx[t]].
        U0: (T, width) start points, width >= the largest n_trainable (a task uses the first n_trainable of
        its row).  Returns the same dict as fit(); rows of "u" are padded with NaN past n_trainable."""
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        sidx = np.ascontiguousarray(spec_idx, dtype=np.int32)
        T = len(cols)
        width = max(sp.n_trainable for sp in self.specs)
        U = np.array(U0, dtype=np.float64, order="C", copy=True)
        if U.ndim != 2 or U.shape[0] != T or U.shape[1] < width or len(sidx) != T:
            raise ValueError(f"U0 must be ({T}, >= {width}) and spec_idx ({T},)")
        opts = opts or lbfgs_opts()
        loss, lml = np.empty(T), np.empty(T)
        nit, nfev, st = (np.empty(T, dtype=np.int32) for _ in range(3))
        ps = None
        if prior_scale is not None:
            ps = np.ascontiguousarray(np.broadcast_to(prior_scale, (T,)), dtype=np.float64)
        _lib.check(self.lib.wv_fit_multi(self.handle, _lib.as_ip(cols), _lib.as_ip(sidx), T, C.byref(opts),
                                         _lib.as_dp(ps) if ps is not None else None, U.shape[1], _lib.as_dp(U),
                                         _lib.as_dp(loss), _lib.as_dp(lml), _lib.as_ip(nit), _lib.as_ip(nfev),
                                         _lib.as_ip(st)), "wv_fit_multi")
        nts = np.array([self.specs[i].n_trainable for i in sidx])
        U[np.arange(U.shape[1])[None, :] >= nts[:, None]] = np.nan
        return {"u": U, "loss": loss, "lml": lml, "nit": nit, "nfev": nfev, "status": st}

    def fit_stream(self, pull, task_args, opts=None, width=None, backlog=None, first_fill=None):
        """Fit a task list that arrives in chunks, keeping the batch full across chunk boundaries: the problem's work
        queue (wv_queue_*) is topped up from `pull()` whenever fewer than `max_batch + backlog` tasks are queued or in
        flight.  `pull()` -> array of task ids (empty: nothing left; e.g. parallel.TaskDealer.pull, which deals chunks
        to the ranks of a job on demand); `task_args(ids)` -> (cols, spec_idx, U0, prior_scale or None) for those ids.
        `first_fill` (parallel.TaskDealer.first_fill): hold no more than this many tasks until the first fit has
        finished - a rank of a small job must not grab more than its share before anything has come back.
        Returns (ids, dict of arrays like fit_multi's) in the order the tasks were drawn.  Fits are bit-identical to
        fit / fit_multi on the same tasks (a slot's arithmetic does not depend on what else is in the batch)."""
        width = int(width or max(sp.n_trainable for sp in self.specs))
        backlog = self.max_batch // 4 if backlog is None else int(backlog)
        hold_full = self.max_batch + max(0, backlog)
        # (the first-fill bound holds until the first fit has come back; afterwards the queue may fill up)
        hold = hold_full if first_fill is None else max(1, min(hold_full, int(first_fill)))
        q = FitQueue(self, opts, width=width)
        drawn, got, spec_of = [], {}, {}
        exhausted = False
        try:
            while True:
                while not exhausted and q.in_system < hold:
                    ids = np.asarray(pull(), dtype=np.int64)
                    if len(ids) == 0:
                        exhausted = True
                        break
                    cols, sidx, U0, ps = task_args(ids)
                    sidx = np.zeros(len(ids), dtype=np.int32) if sidx is None else np.asarray(sidx, dtype=np.int32)
                    q.submit(cols, sidx, U0, ids, prior_scale=ps)
                    drawn.append(ids)
                    spec_of.update(zip(ids.tolist(), sidx.tolist()))
                if q.in_system == 0 and q.done == 0:
                    if exhausted:
                        break
                    continue
                done, _ = q.advance()
                if done:
                    hold = hold_full
                    r = q.fetch(max(done, 64))
                    for j, tg in enumerate(r["tag"]):
                        got[int(tg)] = (r["u"][j].copy(), r["loss"][j], r["lml"][j], r["nit"][j], r["nfev"][j],
                                        r["status"][j])
        finally:
            q.close()
        ids = np.concatenate(drawn) if drawn else np.zeros(0, dtype=np.int64)
        T = len(ids)
        out = {"u": np.full((T, width), np.nan), "loss": np.empty(T), "lml": np.empty(T),
               "nit": np.empty(T, dtype=np.int32), "nfev": np.empty(T, dtype=np.int32),
               "status": np.empty(T, dtype=np.int32)}
        for row, t in enumerate(ids):
            u, out["loss"][row], out["lml"][row], out["nit"][row], out["nfev"][row], out["status"][row] = got[int(t)]
            nt = self.specs[spec_of[int(t)]].n_trainable
            out["u"][row, :nt] = u[:nt]   # (NaN past a structure's own parameters, like fit_multi)
        return ids, out

    def predict(self, col, theta, include_noise=True, Xnew=None):
        """Posterior mean / variance (predict_y when include_noise, else predict_f) at the training inputs
        (Xnew=None) or at an (M, D) array of new inputs."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if Xnew is None:
            M, xp = self.N, None
        else:
            Xnew = np.ascontiguousarray(Xnew, dtype=np.float64)
            if Xnew.ndim != 2 or Xnew.shape[1] != self.D:
                raise ValueError(f"Xnew must be (M, {self.D})")
            M, xp = Xnew.shape[0], _lib.as_dp(Xnew)
        mean, var = np.empty(M), np.empty(M)
        _lib.check(self.lib.wv_predict(self.handle, int(col), _lib.as_dp(theta), xp, M if Xnew is not None else 0,
                                       int(bool(include_noise)), _lib.as_dp(mean), _lib.as_dp(var)), "wv_predict")
        return mean, var

    def whitened_means(self, col, theta, term_masks, jitter=1e-6):
        """Training-input means of the whitened SVGP with the kernel restricted to the additive terms in
        each mask while q_mu is kept (what utilities.py:614-707 computes by popping a kernel)."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        masks = np.ascontiguousarray(term_masks, dtype=np.uint64)
        out = np.empty((len(masks), self.N))
        _lib.check(self.lib.wv_whitened_means(self.handle, int(col), _lib.as_dp(theta),
                                              masks.ctypes.data_as(C.POINTER(C.c_uint64)), len(masks),
                                              float(jitter), _lib.as_dp(out)), "wv_whitened_means")
        return out

    def whitened_state(self, col, theta, jitter=1e-6, want_sqrt=True):
        """(q_mu [N], q_sqrt [N, N] lower or None): the optimum of the reference's whitened variational state at theta
        (model_classes.py:99-100, 149-155) - q_mu = chol(K + jitter I)' alpha, q_sqrt = chol((I + L'L / sigma^2)^-1)."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        q_mu = np.empty(self.N)
        q_sqrt = np.empty((self.N, self.N)) if want_sqrt else None
        _lib.check(self.lib.wv_whitened_state(self.handle, int(col), _lib.as_dp(theta), float(jitter), _lib.as_dp(q_mu),
                                              _lib.as_dp(q_sqrt) if want_sqrt else None), "wv_whitened_state")
        return q_mu, q_sqrt

    # ---- timing ----
    STAGES = ("gram_assembly", "potrf_diag", "chol_trsm", "chol_panel_update", "chol_trailing_syrk",
              "trtri_tri_gemm", "trtri_diag_gemm", "lauum_gemm", "solve_matvec", "grad_trace", "reduce",
              "lauum_gemm_few_fits")

    def timing(self, on, stages=None):
        """on=False: off; on=True: all stages, or only the named `stages`."""
        code = 0
        if on:
            code = 1
            if stages:
                code = 0
                for s in stages:
                    code |= 1 << (self.STAGES.index(s) + 1)
        self.lib.wv_stage_timing_enable(self.handle, code)

    def timing_reset(self):
        self.lib.wv_stage_times_reset(self.handle)

    def stage_times(self):
        cap = 32
        names = (C.c_char_p * cap)()
        ms, fl, by = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        ln = np.zeros(cap, dtype=np.int64)
        n = self.lib.wv_stage_times(self.handle, cap, names, _lib.as_dp(ms),
                                    ln.ctypes.data_as(C.POINTER(C.c_int64)), _lib.as_dp(fl), _lib.as_dp(by))
        return {names[i].decode(): {"ms": float(ms[i]), "launches": int(ln[i]), "flops": float(fl[i]),
                                    "bytes": float(by[i])} for i in range(n)}

    # ---- bring-up ----
    def dbg_stage(self, col, theta, what):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        npad = int(self.lib.wv_dbg_npad(self.handle))
        out = np.empty(npad if what >= 3 else (npad, npad))
        _lib.check(self.lib.wv_dbg_stage(self.handle, int(col), _lib.as_dp(theta), int(what), _lib.as_dp(out)),
                   "wv_dbg_stage")
        return out


class FitQueue:
    """The fit work queue of a (multi-structure) Problem, fed incrementally (wv_queue_*): `submit` tasks at any time,
    `advance` runs one scheduling step, `fetch` returns finished fits.  Fits equal Problem.fit_multi's bit for bit."""

    def __init__(self, prob: Problem, opts=None, width: Optional[int] = None):
        self.prob, self.lib = prob, prob.lib
        self.opts = opts or lbfgs_opts()
        self.width = int(width or max(sp.n_trainable for sp in prob.specs))
        self.handle = C.c_void_p()
        _lib.check(self.lib.wv_queue_create(prob.handle, C.byref(self.opts), C.byref(self.handle)), "wv_queue_create")
        self.done = self.in_system = 0

    def submit(self, cols, spec_idx, U0, tags, prior_scale=None):
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        sidx = np.ascontiguousarray(spec_idx, dtype=np.int32)
        tags = np.ascontiguousarray(tags, dtype=np.int64)
        U = np.ascontiguousarray(np.atleast_2d(U0), dtype=np.float64)
        T = len(cols)
        if U.shape[0] != T or len(sidx) != T or len(tags) != T:
            raise ValueError("cols, spec_idx, U0 and tags disagree on the number of tasks")
        ps = None
        if prior_scale is not None:
            ps = np.ascontiguousarray(np.broadcast_to(prior_scale, (T,)), dtype=np.float64)
        _lib.check(self.lib.wv_queue_submit(self.handle, T, _lib.as_ip(cols), _lib.as_ip(sidx),
                                            _lib.as_dp(ps) if ps is not None else None, U.shape[1], _lib.as_dp(U),
                                            tags.ctypes.data_as(C.POINTER(C.c_int64))), "wv_queue_submit")
        self.in_system += T

    def advance(self):
        """One scheduling step; returns (finished fits waiting, tasks queued or in flight)."""
        d, n = C.c_int32(), C.c_int32()
        _lib.check(self.lib.wv_queue_advance(self.handle, C.byref(d), C.byref(n)), "wv_queue_advance")
        self.done, self.in_system = int(d.value), int(n.value)
        return self.done, self.in_system

    def fetch(self, cap: int = 4096):
        """dict of arrays (tag, u, loss, lml, nit, nfev, status) for up to `cap` finished fits, oldest first."""
        w = self.width
        tags = np.empty(cap, dtype=np.int64)
        U = np.empty((cap, w))
        loss, lml = np.empty(cap), np.empty(cap)
        nit, nfev, st = (np.empty(cap, dtype=np.int32) for _ in range(3))
        n = C.c_int32()
        _lib.check(self.lib.wv_queue_fetch(self.handle, cap, w, tags.ctypes.data_as(C.POINTER(C.c_int64)),
                                           _lib.as_dp(U), _lib.as_dp(loss), _lib.as_dp(lml), _lib.as_ip(nit),
                                           _lib.as_ip(nfev), _lib.as_ip(st), C.byref(n)), "wv_queue_fetch")
        k = int(n.value)
        self.done = max(0, self.done - k)
        return {"tag": tags[:k], "u": U[:k], "loss": loss[:k], "lml": lml[:k], "nit": nit[:k], "nfev": nfev[:k],
                "status": st[:k]}

    def close(self):
        if self.handle:
            self.lib.wv_queue_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
