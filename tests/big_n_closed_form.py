"""By-hand check at sizes beyond the unit tests (N = 16384 is the largest there): a kernel whose exact LML has a closed
form at any N - Constant(v) + Categorical(w)[two levels] + noise s, zero mean - so no O(N^3) oracle is needed:
K_y = v 11' + w (1_A 1_A' + 1_B 1_B') + s I is s I plus a rank-2 matrix; log|K_y| and the quadratic form follow from the
2 x 2 capacitance matrix (matrix determinant lemma / Woodbury).  Value-only path, value + gradient path (gradient
against central differences of the closed form).     python tests/big_n_closed_form.py [N=32768]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from waveome_amd import ModelSpec, kernels as K  # noqa: E402
from waveome_amd.engine import Context, Problem  # noqa: E402


from oracle.closed_forms import (closed_form, closed_form_lin_poly, closed_form_periodic,  # noqa: E402,F401
                                 headline_closed_form, low_rank_lml, periodic_features, se_features)


def main_headline(N, dtype="float64"):
    """The headline structure itself, one fit, at a size far beyond the bench's: device LML against the closed form
    (`... N headline float32`: on a float32 context, against the float32 parity tolerance 1e-6)."""
    import bench

    saved = bench.N_HEAD
    bench.N_HEAD = N
    try:
        X, Y = bench.make_inputs(1)
    finally:
        bench.N_HEAD = saved
    ms = bench.headline_spec()
    ctx = Context(0, dtype=dtype)
    tol = 1e-9 if dtype == "float64" else 1e-6
    prob = Problem(ctx, X, Y.T, ms, max_batch=1)
    t0 = time.time()
    want = headline_closed_form(ms.theta(), X, Y[0])
    tc = time.time() - t0
    t0 = time.time()
    lml, st = prob.lml([0], ms.theta()[None, :])
    t1 = time.time() - t0
    t0 = time.time()
    f, g, st2 = prob.value_grad([0], ms.unconstrain()[None, :])
    t2 = time.time() - t0
    prob.close()
    ctx.close()
    rel = max(abs(lml[0] - want), abs(-f[0] - want)) / abs(want)
    print(f"headline structure, {dtype}, N = {N}: closed form {want:.6f} ({tc:.1f} s on the CPU), device value-only {lml[0]:.6f} ({t1:.2f} s), "
          f"value + gradient {-f[0]:.6f} ({t2:.2f} s = {N ** 3 / t2 / 1e12:.1f} TFLOP/s on N^3); worst rel {rel:.1e}, status {st[0]} {st2[0]}")
    print("# ok" if rel < tol else "# MISMATCH")
    return 0 if rel < tol else 1


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    if len(sys.argv) > 2 and sys.argv[2] == "headline":
        return main_headline(N, sys.argv[3] if len(sys.argv) > 3 else "float64")
    rng = np.random.default_rng(1)
    lab = rng.integers(0, 2, N).astype(float)
    X = np.column_stack([rng.normal(size=N), lab])
    y = 0.7 + 0.5 * (lab == 1) + 0.3 * rng.normal(size=N)
    v, w, s = 0.8, 0.4, 0.09
    kern = K.Sum([K.Constant(variance=v), K.Categorical(active_dims=[1], variance=w)])
    ms = ModelSpec(kern, 2, K.GaussianLikelihood(variance=s), mean_function=None)
    ctx = Context(0)
    t0 = time.time()
    prob = Problem(ctx, X, y[:, None], ms, max_batch=1)
    print(f"N = {N}: problem created in {time.time() - t0:.1f} s, workspace {prob.workspace_bytes / 2**30:.1f} GiB", flush=True)
    th = ms.theta()
    ref = closed_form(v, w, s, lab, y)
    t0 = time.time()
    lml, st = prob.lml([0], th[None, :])
    print(f"  value-only path: LML {lml[0]:.6f} vs closed form {ref:.6f}  (rel {abs(lml[0] - ref) / abs(ref):.1e}), status {st[0]}, "
          f"{time.time() - t0:.2f} s", flush=True)
    t0 = time.time()
    f, g, st = prob.value_grad([0], ms.unconstrain()[None, :])
    dt = time.time() - t0
    print(f"  value + gradient: loss {f[0]:.6f} (rel {abs(-f[0] - ref) / abs(ref):.1e}), {dt:.2f} s = {N ** 3 / dt / 1e12:.1f} TFLOP/s on N^3", flush=True)
    # gradient in the unconstrained coordinates against central differences of the closed form
    u = ms.unconstrain()
    num = np.zeros_like(u)
    for k in range(len(u)):
        h = 1e-5
        up, um = u.copy(), u.copy()
        up[k] += h
        um[k] -= h
        tp, tm = ms.constrain(up), ms.constrain(um)
        num[k] = -(closed_form(tp[0], tp[1], tp[2], lab, y) - closed_form(tm[0], tm[1], tm[2], lab, y)) / (2 * h)
    print(f"  gradient {g[0]} vs central differences {num}: max rel {np.max(np.abs(g[0] - num) / np.maximum(1.0, np.abs(num))):.1e}")
    prob.close()
    ctx.close()
    ok = abs(lml[0] - ref) / abs(ref) < 1e-9 and abs(-f[0] - ref) / abs(ref) < 1e-9 and \
        np.max(np.abs(g[0] - num) / np.maximum(1.0, np.abs(num))) < 1e-5
    print("# ok" if ok else "# MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
