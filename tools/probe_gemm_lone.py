"""What one workgroup per CU sustains (no partner wave on its SIMDs): the GEMM kernels with extra dynamic LDS
(WV_GEMM_LDS_PAD, diagnostic build) against the normal two per CU, both K loops.  If a lone workgroup ran at the full
MFMA rate the one-per-CU figure would equal the two-per-CU one; what is missing is what the partner's prologue /
epilogue costs the other workgroup."""
import os
import sys

sys.path.insert(0, ".")
import numpy as np

import _probes_lib  # noqa: F401
from waveome_amd.engine import Context

c = Context(0)
print(f"{'M':>5} {'K':>5} {'nb':>4} mode | TF: asm 2/CU, asm 1/CU, hipcc 2/CU, hipcc 1/CU")
for (M, N, K, nb, mode) in [(2048, 2048, 2048, 64, 0), (1024, 1024, 1024, 256, 0), (512, 512, 512, 512, 0), (2048, 2048, 2048, 256, 1)]:
    res = {}
    for r in range(3):
        for a in (1, 0):
            for pad in (0, 40000):
                os.environ["WV_GEMM_ASM"] = str(a)
                os.environ["WV_GEMM_LDS_PAD"] = str(pad)
                tf, ms = c.gemm_bench(M, N, K, False, False, 128 + 1000 * mode, nb, 3)
                res.setdefault((a, pad), []).append(tf)
    print(f"{M:5d} {K:5d} {nb:4d} {mode:4d} | " + "  ".join(f"{np.median(res[k]):6.2f}" for k in [(1, 0), (1, 40000), (0, 0), (0, 40000)]), flush=True)
