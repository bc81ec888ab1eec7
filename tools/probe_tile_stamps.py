"""Where one 128 x 128 tile of the float64 GEMM spends its cycles inside a full launch (diagnostic build): shader-clock
stamps of one workgroup in the middle of the grid - entry, first chunk in LDS, end of the K loop, the two halves of the
epilogue (staged / stored) - with two workgroups per CU and with one (WV_GEMM_LDS_PAD), both K loops."""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
import numpy as np

import _probes_lib  # noqa: F401
from waveome_amd.engine import Context

c = Context(0)
for (M, K, nb, mode, name) in [(2048, 2048, 64, 0, "full 2048^3 x 64"), (1024, 1024, 256, 0, "full 1024^3 x 256"), (1024, 1024, 512, 2, "update 1024 x 512 (beta = 1)")]:
    for a, dbg in ((1, 0), (1, 128), (0, 0)):   # (WV_GEMM_DBG bit 128: no raised priority outside the K loop)
        for pad in (0, 40000):
            os.environ["WV_GEMM_ASM"] = str(a)
            os.environ["WV_GEMM_DBG"] = str(dbg)
            os.environ["WV_GEMM_LDS_PAD"] = str(pad)
            tf, ms = c.gemm_bench(M, M, K, False, False, 128 + 1000 * mode, nb, 2)
            out = (C.c_uint64 * 8)()
            assert c.lib.wv_dbg_gemm_stamps(out) == 0
            s = np.array(list(out), dtype=np.int64)
            print(f"{name:28s} {('asm  ' if not dbg else 'asm-p') if a else 'hipcc'} {'1/CU' if pad else '2/CU'} {tf:6.1f} TF | cycles: prologue {s[1] - s[0]:6d}  K loop {s[2] - s[1]:8d} "
                  f"({(s[2] - s[1]) / (K // 16):6.0f}/chunk)  epilogue: stage0 {s[4] - s[2]:6d} store0 {s[5] - s[4]:6d} stage1 {s[6] - s[5]:6d} store1 {s[7] - s[6]:6d}  total {s[3] - s[0]:8d}", flush=True)
