"""An oracle-independent exact answer for the Matern family: a Matern-(p + 1/2) GP on the real line is the stationary
solution of a linear stochastic differential equation of order p + 1 (Whittle 1954; Hartikainen & Sarkka 2010), so the
log marginal likelihood of noisy observations at sorted 1-D inputs is what a Kalman filter accumulates - O(N), no Gram
matrix, no Cholesky factor, nothing shared with oracle/gp_oracle.py or the device kernels except the definition of the
process.  Used by tests/test_oracle.py and tests/test_gpu_parity.py: it pins Matern-3/2, for which the reference holds
no stored value, through a derivation that never evaluates (1 + sqrt(3) r) exp(-sqrt(3) r)."""
import numpy as np
from scipy.linalg import expm


def _model(kind, variance, lengthscale):
    if kind == "m12":
        lam = 1.0 / lengthscale
        return np.array([[-lam]]), np.array([[variance]])
    if kind == "m32":
        lam = np.sqrt(3.0) / lengthscale
        F = np.array([[0.0, 1.0], [-lam ** 2, -2.0 * lam]])
        return F, np.diag([variance, lam ** 2 * variance])
    if kind == "m52":
        lam = np.sqrt(5.0) / lengthscale
        F = np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [-lam ** 3, -3.0 * lam ** 2, -3.0 * lam]])
        kap = variance * lam ** 2 / 3.0
        Pinf = np.array([[variance, 0.0, -kap], [0.0, kap, 0.0], [-kap, 0.0, lam ** 4 * variance]])
        return F, Pinf
    raise ValueError(kind)


def kalman_lml(kind, variance, lengthscale, noise, x, y, mean=0.0):
    """log N(y | mean, K_matern(x, x) + noise I) for 1-D inputs x (any order, distinct values)."""
    order = np.argsort(x)
    x, y = np.asarray(x, float)[order], np.asarray(y, float)[order] - mean
    F, Pinf = _model(kind, variance, lengthscale)
    d = F.shape[0]
    m, P = np.zeros(d), Pinf.copy()
    ll = 0.0
    for k in range(len(x)):
        if k > 0:
            A = expm(F * (x[k] - x[k - 1]))
            m = A @ m
            P = A @ P @ A.T + (Pinf - A @ Pinf @ A.T)
        S = P[0, 0] + noise
        v = y[k] - m[0]
        ll += -0.5 * (np.log(2.0 * np.pi * S) + v * v / S)
        gain = P[:, 0] / S
        m = m + gain * v
        P = P - np.outer(gain, P[0, :])
        P = 0.5 * (P + P.T)
    return ll
