"""Where a small-tile GEMM launch spends its time with few workgroups in flight (the products of one fit's evaluation):
shader-clock stamps of workgroup 0 at entry / first chunk in LDS / end of the K loop / end (probes build)."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _probes_lib  # noqa: F401,E402
import numpy as np  # noqa: E402
from waveome_amd.engine import Context  # noqa: E402

c = Context(0)
import os
print("WV_GEMM_DBG =", os.environ.get("WV_GEMM_DBG", "0"), "(4 no requests, 8 no MFMAs, 16 no barrier, 64 no deposits; results wrong when set)")
for (M, N, K, akc, name) in [(128, 128, 128, 0, "128^3 FF"), (256, 256, 256, 0, "256^3 FF"),
                             (512, 512, 512, 0, "512^3 FF"), (1024, 1024, 1024, 0, "1024^3 FF")]:
    tf, ms = c.gemm_bench(M, N, K, bool(akc), False, 32, 1, 5)
    out = (C.c_uint64 * 8)()
    assert c.lib.wv_dbg_gemm_stamps(out) == 0
    s = np.array(list(out), dtype=np.int64)
    nch = K // 16
    print(f"{name:14s} {ms * 1e3:7.1f} us per launch | cycles: prologue {s[1] - s[0]:6d}  K loop {s[2] - s[1]:7d} "
          f"({(s[2] - s[1]) / nch:6.0f} per chunk of 16)  epilogue {s[3] - s[2]:6d}  total {s[3] - s[0]:7d}", flush=True)
