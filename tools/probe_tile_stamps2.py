"""Shader-clock stamps of one 128 x 128 tile in the middle of a launch (diagnostic build), short contractions and the
triangular forms: entry / first chunk staged / end of K loop / epilogue phases.  python3 tools/probe_tile_stamps2.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
import numpy as np

import _probes_lib  # noqa: F401
from waveome_amd.engine import Context

c = Context(0)
for (M, K, nb, akc, mode, name) in [(1024, 1024, 256, 0, 0, "full 1024^3 x 256"), (512, 512, 512, 0, 0, "full 512^3 x 512"),
                                    (256, 256, 1024, 0, 0, "full 256^3 x 1024"), (1024, 1024, 512, 1, 3, "S-like 1024 x 512"),
                                    (1024, 1024, 512, 0, 6, "M21-like 1024 x 512"), (2048, 2048, 512, 0, 1, "M'M 2048 x 512")]:
    for pad in (0, 40000):
        os.environ["WV_GEMM_ASM"] = "1"
        os.environ["WV_GEMM_DBG"] = "0"
        os.environ["WV_GEMM_LDS_PAD"] = str(pad)
        tf, ms = c.gemm_bench(M, M, K, bool(akc), False, 128 + 1000 * mode, nb, 2)
        out = (C.c_uint64 * 8)()
        assert c.lib.wv_dbg_gemm_stamps(out) == 0
        s = np.array(list(out), dtype=np.int64)
        print(f"{name:24s} {'1/CU' if pad else '2/CU'} {tf:6.1f} TF | cycles: prologue {s[1] - s[0]:6d}  K loop {s[2] - s[1]:8d}  "
              f"epilogue: stage0 {s[4] - s[2]:6d} store0 {s[5] - s[4]:6d} stage1 {s[6] - s[5]:6d} store1 {s[7] - s[6]:6d}  total {s[3] - s[0]:8d}", flush=True)
