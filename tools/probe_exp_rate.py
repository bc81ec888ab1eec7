"""exp2_64 (wv_gram.hip) against its candidate replacements, in the diagnostic build (make probes):
rate of each variant at 2 / 3 / 4 waves per SIMD, and bit identity against the shipped one on random arguments.
usage (GPU box): python3 tools/probe_exp_rate.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "libwv_probes.so"))
lib.wv_dbg_exp_rate.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_double)]
lib.wv_dbg_exp_check.argtypes = [C.c_int32, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double)]

names = {0: "shipped", 1: "3-address fma, SGPR coefficients", 2: "+ shift rint/convert", 3: "+ exponent add for ldexp"}
print("variant                              waves/SIMD   Gexp/s   cycles per exp and SIMD (2.4 GHz)")
for v in range(4):
    for w in (2, 3, 4):
        out = C.c_double(0)
        rc = lib.wv_dbg_exp_rate(v, 4000, w, -3.0, 0.37, C.byref(out))
        assert rc == 0, rc
        cyc = 2.4e9 * 1024 / (out.value * 1e9 / 64)
        print(f"{v} {names[v]:34s} {w:6d} {out.value:10.1f} {cyc:10.1f}")
rng = np.random.default_rng(0)
n = 1 << 24
sets = {"uniform [-2000, 0]": -2000.0 * rng.random(n), "uniform [-65000, 0]": -65000.0 * rng.random(n),
        "near zero": -rng.random(n) * 2.0, "half-way cases": -(rng.integers(0, 60000, n) + 0.5),
        "integers": -rng.integers(0, 65000, n).astype(np.float64)}
for v in (1, 2, 3):
    for name, t in sets.items():
        t = np.ascontiguousarray(t, dtype=np.float64)
        nb = C.c_int64(0)
        fb = (C.c_double * 3)()
        rc = lib.wv_dbg_exp_check(v, t.ctypes.data_as(C.POINTER(C.c_double)), n, C.byref(nb), fb)
        assert rc == 0, rc
        print(f"variant {v} vs shipped, {name:22s}: {nb.value} of {n} differ", ("" if nb.value == 0 else f"(first: t={fb[0]!r} {fb[1]!r} {fb[2]!r})"))
