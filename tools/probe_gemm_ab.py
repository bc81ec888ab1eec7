"""A/B timing of the float64 GEMM's K loop in ONE process (diagnostic build): the hand-scheduled instruction stream
(wv_gemm_kloop.inc, WV_GEMM_ASM=1, the release kernels) against hipcc's loop (WV_GEMM_ASM=0, TAG bit 5), interleaved
rounds, median and best.  Modes: 0 full, 1 K_y^-1 = M'M (lower tiles + diagonal launch, contraction from the tile row),
2 trailing update (lower tiles, beta = 1), 3 contraction trimmed to k < n0 + BN, 6 trimmed to k < m0 + BM."""
import os
import sys

sys.path.insert(0, ".")
import numpy as np

import _probes_lib  # noqa: F401
from waveome_amd.engine import Context

c = Context(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cases = [(2048, 2048, 2048, 64, 0, 0, 0), (1024, 1024, 1024, 256, 0, 0, 0), (1024, 1024, 1024, 256, 1, 0, 0),
         (2048, 2048, 2048, 256, 0, 0, 1), (2048, 2048, 2048, 512, 0, 0, 1), (1024, 1024, 1024, 512, 0, 0, 2),
         (1024, 1024, 1024, 512, 1, 0, 3), (1024, 1024, 1024, 512, 0, 0, 6), (512, 512, 512, 512, 0, 0, 0),
         (512, 512, 512, 512, 0, 0, 2), (256, 256, 256, 1024, 0, 0, 0)]
print(f"{'M':>5} {'K':>5} {'nb':>4} akc mode | asm TF (median, best) | hipcc TF (median, best) | asm / hipcc | asm without raised priority outside the K loop")
for (M, N, K, nb, akc, bkc, mode) in cases:
    res = {1: [], 0: [], 2: []}
    for r in range(rounds):
        for a in (1, 0, 2):
            os.environ["WV_GEMM_ASM"] = "0" if a == 0 else "1"
            os.environ["WV_GEMM_DBG"] = "128" if a == 2 else "0"
            tf, ms = c.gemm_bench(M, N, K, bool(akc), bool(bkc), 128 + 1000 * mode, nb, 3)
            res[a].append(tf)
    m1, m0 = np.median(res[1]), np.median(res[0])
    print(f"{M:5d} {K:5d} {nb:4d} {akc:3d} {mode:4d} | {m1:7.2f} {max(res[1]):7.2f}     | {m0:7.2f} {max(res[0]):7.2f}       | {m1 / m0:6.3f} | {np.median(res[2]):7.2f}",
          flush=True)
