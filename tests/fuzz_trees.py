"""Differential fuzz of the device value / gradient / Gram path against the oracle - a longer run of what
tests/test_gpu_parity.py::test_random_kernel_trees_match_oracle does for ten seeds: seeded random Sum / Product trees
over every kernel type, random sizes (padding edges included), random parameter points, Horseshoe priors, several
fits per call.  Not collected by pytest (no test_ prefix); run by hand on a GPU box:

    python tests/fuzz_trees.py [first_seed=1000] [n_seeds=300] [float64|float32] > profiles/rNN_fuzz_trees.txt

(float32: the float32 context against the same float64 oracle with the float32 parity tolerances - loss / LML 1e-6,
gradient 1e-4; the Gram matrix is then one float32 rounding of each entry.)

Prints one line per seed and a summary; exits non-zero on a mismatch."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import test_gpu_parity as T  # noqa: E402  (the tree generator and data of the unit test)
from oracle import gp_oracle as o  # noqa: E402
from waveome_amd import ModelSpec, kernels as K  # noqa: E402
from waveome_amd.engine import Context, Problem  # noqa: E402


def one(ctx, seed):
    rng = np.random.default_rng(seed)
    kern = T._random_tree(rng, int(rng.integers(1, 4)))
    ms = ModelSpec(kern, 4, K.GaussianLikelihood(), K.ConstantMean() if rng.random() < 0.7 else None)
    if rng.random() < 0.6:
        T._with_priors(ms, scale=float(rng.choice([0.1, 1.0, 10.0])))
    if sum(len(t) for t in ms.terms) > 128 or len(ms.terms) > 64 or max(len(t) for t in ms.terms) > 4:
        return "skipped (beyond the limits of one device spec)", 0.0, 0.0, 0.0
    N = int(rng.choice([1, 2, 17, 63, 64, 65, 127, 128, 129, 191, 200, 257, 333, 400]))
    X, Y = T._data(seed, N)
    nb = int(rng.integers(1, 4))
    prob = Problem(ctx, X, Y, ms, max_batch=nb)
    d = ms.to_dict()
    th = ms.theta()
    Kg = prob.gram(th)
    Ko = o.gram(d, th, X) + th[ms.noise] * np.eye(N)
    gram_err = float(np.max(np.abs(Kg - Ko)) / max(1.0, np.max(np.abs(Ko))))
    U = np.stack([ms.unconstrain() + float(rng.choice([0.2, 1.0])) * rng.normal(size=ms.n_trainable)
                  for _ in range(nb)])
    cols = rng.integers(0, Y.shape[1], nb).astype(np.int32)
    loss, grad, st = prob.value_grad(cols, U)
    thetas = np.stack([ms.constrain(u) for u in U])
    lml, st2 = prob.lml(cols, thetas)
    prob.close()
    loss_err = grad_err = 0.0
    note = ""
    for b in range(nb):
        fo, go = o.objective(d, U[b], X, Y[:, cols[b]])
        if not np.isfinite(fo):
            if st[b] == 0 and np.isfinite(loss[b]):
                note += " ORACLE-INF-DEVICE-FINITE"
            continue
        if st[b] != 0 or not np.isfinite(loss[b]):
            note += f" DEVICE-FAILED(status {st[b]})-ORACLE-FINITE({fo:.6g})"
            continue
        loss_err = max(loss_err, abs(loss[b] - fo) / max(1.0, abs(fo)))
        grad_err = max(grad_err, float(np.max(np.abs(grad[b] - go)) / max(1.0, np.max(np.abs(go)))))
        lo = o.lml(d, thetas[b], X, Y[:, cols[b]])
        loss_err = max(loss_err, abs(lml[b] - lo) / max(1.0, abs(lo)))
    return f"N={N} fits={nb} terms={len(ms.terms)} params={ms.n_trainable}{note}", gram_err, loss_err, grad_err


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    dtype = sys.argv[3] if len(sys.argv) > 3 else "float64"
    lim = (1e-12, 1e-9, 1e-6) if dtype == "float64" else (2e-7, 1e-6, 1e-4)
    ctx = Context(0, dtype=dtype)
    worst = [0.0, 0.0, 0.0]
    bad = 0
    t0 = time.time()
    for seed in range(first, first + n):
        what, ge, le, gre = one(ctx, seed)
        flag = ""
        if ge > lim[0] or le > lim[1] or gre > lim[2] or "DEVICE" in what or "ORACLE-INF" in what:
            flag = "  <-- MISMATCH"
            bad += 1
        worst = [max(a, b) for a, b in zip(worst, (ge, le, gre))]
        print(f"seed {seed}: {what}; gram {ge:.1e} loss {le:.1e} grad {gre:.1e}{flag}", flush=True)
    print(f"# {dtype}: {n} seeds in {time.time() - t0:.0f} s: worst gram {worst[0]:.1e} (limit {lim[0]:g}), loss / LML {worst[1]:.1e} "
          f"({lim[1]:g}), gradient {worst[2]:.1e} ({lim[2]:g}); mismatches: {bad}")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
