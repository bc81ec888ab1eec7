"""Where the fused 128-node kernel (potrf128) spends its time with ONE workgroup in flight: shader-clock stamps of
thread 0 at the phase boundaries.  Needs the diagnostic build:  make -C waveome_amd/csrc stamps
  python3 tools/probe_potrf_stamps.py [N]      (loads tools/libwv_stamps.so through WAVEOME_AMD_LIB)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WAVEOME_AMD_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwv_stamps.so"))
import numpy as np
import bench
from waveome_amd.engine import Context, Problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bench.N_HEAD = N
X, Y = bench.make_inputs(8)
spec = bench.headline_spec()
prob = Problem(Context(0), X, Y.T, spec, max_batch=1)
cols = np.zeros(1, dtype=np.int32)
U = spec.unconstrain()[None, :]
for _ in range(3):
    prob.value_grad(cols, U)
lib = ctypes.CDLL(os.environ["WAVEOME_AMD_LIB"])
out = (ctypes.c_ulonglong * 32)()
assert lib.wv_dbg_stamps(out) == 0
s = np.array(list(out), dtype=np.int64)
names = {1: "load A11", 2: "stage + factor64c (A11) + clear X", 3: "16x16 diagonal inverses (wave 0) beside the log-det (wave 3)", 4: "invert64 (M11)", 5: "write M11 / M11'",
         6: "S = A21 M11' (+ staging)", 7: "A22 -= S S', T2 = S M11", 8: "stage + factor64c (Schur) + clear X", 9: "16x16 diagonal inverses beside the log-det",
         10: "invert64 (M22)", 11: "write M22 / M22'", 12: "M21 = -M22 T2, stores"}
tot = s[12] - s[0]
print(f"potrf128, one workgroup, last node of an evaluation at N = {N}: {tot} shader-clock ticks")
for i in range(1, 13):
    print(f"  {names[i]:32s} {s[i] - s[i - 1]:8d}  {100.0 * (s[i] - s[i - 1]) / tot:5.1f} %")
print("  inside the second invert64 (after the one-wave 16 x 16 diagonal inverses): level 16: %d, level 32: %d" % (s[19] - s[18], s[10] - s[19]))
prob.close()
