"""Accuracy of the 128-node leaf on an ill-conditioned block (the case of tests/test_gpu_parity.py::
test_model_with_noise_on_its_floor_predicts: N = 90, cond(K_y) = 2.7e7): device LML against the LAPACK oracle and an
80-bit reference (by-hand Cholesky in numpy longdouble).  Run once per library (WAVEOME_AMD_LIB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import gp_oracle as o
from waveome_amd import kernels as K, ModelSpec
from waveome_amd.engine import Context, Problem

src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_gpu_parity.py")).read()
i = src.index("def _data(")
ns = {"np": np}
exec(src[i:src.index("\n\n", i)], ns)
ld = np.longdouble
ctx = Context(0)
for seed, n, ell, noise in ((5, 90, 0.05, 1e-6), (6, 128, 0.05, 1e-6), (7, 120, 0.3, 1e-6), (8, 128, 1.0, 1e-4)):
    X, Y = ns["_data"](seed, n)
    kern = K.Sum([K.SquaredExponential(active_dims=[0], variance=2.0, lengthscales=ell), K.Categorical(active_dims=[3], variance=0.5)])
    ms = ModelSpec(kern, X.shape[1], K.GaussianLikelihood(variance=noise), K.ConstantMean())
    th, d = ms.theta(), ms.to_dict()
    Ky = (o.gram(d, th, X) + th[ms.noise] * np.eye(n)).astype(ld)
    r = o._resid(d, th, Y[:, 0]).astype(ld)
    Lc = np.zeros((n, n), dtype=ld)
    for j in range(n):
        Lc[j, j] = np.sqrt(Ky[j, j] - np.dot(Lc[j, :j], Lc[j, :j]))
        for q in range(j + 1, n):
            Lc[q, j] = (Ky[q, j] - np.dot(Lc[q, :j], Lc[j, :j])) / Lc[j, j]
    z = np.zeros(n, dtype=ld)
    for q in range(n):
        z[q] = (r[q] - np.dot(Lc[q, :q], z[:q])) / Lc[q, q]
    ref = -ld(0.5) * np.dot(z, z) - np.sum(np.log(np.diag(Lc))) - ld(0.5) * n * np.log(ld(2) * np.pi)
    prob = Problem(ctx, X, Y[:, :1], ms, max_batch=1)
    dev, st = prob.lml([0], th[None, :])
    prob.close()
    lap = o.lml(d, th, X, Y[:, 0])
    cond = np.linalg.cond(Ky.astype(np.float64))
    print(f"{os.environ.get('WAVEOME_AMD_LIB', 'in-tree')[-28:]:>28} N={n} cond {cond:.1e}: device {dev[0]:.10f} rel err {float(abs((ld(dev[0]) - ref) / ref)):.2e}; "
          f"LAPACK rel err {float(abs((ld(lap) - ref) / ref)):.2e}; cond*eps {cond * 1.1e-16:.1e}")
