"""Writes tests/golden/kat.json: known-answer vectors pinned by the reference's own
stored notebook outputs (the reference has no test suite).

Inputs are regenerated from the notebooks' seeded NumPy legacy RNG calls / inline
literals; expected values are the numbers printed in the stored cell outputs:

* KAT-1  kernel_learning/gpflow_basic_colab.ipynb cells 3,4,6,7
* KAT-2  kernel_learning/gpflow_basic.ipynb cells 28-31
* KAT-3  kernel_learning/gpflow_basic_colab.ipynb cell 7 (comment "M12 = -10.68")

Run:  python tests/golden/make_kat.py
"""
import json
import os

import numpy as np

np.random.seed(9102)
X1 = np.random.uniform(low=0, high=10, size=10)
Y1 = np.sin(X1) + np.random.normal(scale=0.5, size=10)

X2 = [0, 0.05, 0.05, 0.15, 0.17, 0.21, 0.61, 0.79, 0.8, 0.9, 0.9, 0.95]
Y2 = [3.6, 3.7, 3.5, 3.1, 3.2, 3.8, 3.5, 3.6, 3.0, 1.8, 1.6, 1.4]

out = {
    "kat1": {
        "source": "kernel_learning/gpflow_basic_colab.ipynb cells 3-7",
        "model": "gpflow.models.GPR(SquaredExponential(), mean_function=None), Scipy().minimize(maxiter=100)",
        "X": X1.tolist(), "Y": Y1.tolist(), "kernel": "se",
        "init": {"variance": 1.0, "lengthscales": 1.0, "noise": 1.0},
        "expected": {"variance": 0.749872, "lengthscales": 0.740194, "noise": 0.0379764,
                     "lml": -9.914289155637},
        "printed_sig_figs": 6,
    },
    "kat2": {
        "source": "kernel_learning/gpflow_basic.ipynb cells 28-31",
        "model": "gpflow.models.GPR(Matern52(), mean_function=None), Scipy().minimize(maxiter=100)",
        "X": X2, "Y": Y2, "kernel": "m52",
        "init": {"variance": 1.0, "lengthscales": 1.0, "noise": 1.0},
        "expected": {"variance": 7.76607, "lengthscales": 0.492535, "noise": 0.0969323},
        "printed_sig_figs": 6,
    },
    "kat3": {
        "source": "kernel_learning/gpflow_basic_colab.ipynb cell 7 comment '# M12 = -10.68'",
        "model": "same data as kat1, Matern12",
        "X": X1.tolist(), "Y": Y1.tolist(), "kernel": "m12",
        "init": {"variance": 1.0, "lengthscales": 1.0, "noise": 1.0},
        "expected": {"lml_2dp": -10.68},
    },
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote", path)
