"""Probes build: does delaying one of the two workgroups that share a CU (by half a tile's time, at the start of a launch
only) break their lockstep?  Uniform-K products, 16 tiles per slot; four guesses at which workgroups share a CU."""
import os, sys
sys.path.insert(0, ".")
import _probes_lib  # noqa: F401
from waveome_amd.engine import Context
c = Context(0)
os.environ["WV_GEMM_ASM"] = "1"; os.environ["WV_GEMM_DBG"] = "0"
for (M, K, nb, name) in ((512, 512, 512, "512^3 x 512"), (1024, 512, 128, "1024^2 K=512 x 128"), (1024, 1024, 128, "1024^3 x 128")):
    row = []
    for st in ("0", "70000,0", "70000,1", "70000,2", "70000,3", "140000,0", "140000,1"):
        os.environ["WV_GEMM_STAGGER"] = st
        best = max(c.gemm_bench(M, M, K, False, False, 128, nb, 4)[0] for _ in range(3))
        row.append(f"{st}: {best:.1f}")
    print(f"{name:22s} TFLOP/s by WV_GEMM_STAGGER=cycles,mode | " + " | ".join(row), flush=True)
