"""Launch time against tiles per workgroup slot at fixed tile shape (probes build): plain K = 512 and K = 1024 products,
nb matrices of 16 / 64 tiles -> nb x tiles / 512 tiles per slot.  The slope is the slot time per tile (incl. whatever
passes between one workgroup's end and the next one's start); compare with the in-kernel stamps of one tile."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import numpy as np
import _probes_lib  # noqa: F401
from waveome_amd.engine import Context
c = Context(0)
os.environ["WV_GEMM_ASM"] = "1"; os.environ["WV_GEMM_DBG"] = "0"
for (M, K) in ((512, 512), (1024, 1024), (1024, 512), (1024, 256)):
    tiles = (M // 128) ** 2
    rows = []
    for nb in (32, 64, 128, 256, 512, 1024):
        if M == 1024 and nb > 512: continue
        tf, ms = c.gemm_bench(M, M, K, False, False, 128, nb, 4)
        out = (C.c_uint64 * 8)(); c.lib.wv_dbg_gemm_stamps(out); s = np.array(list(out), dtype=np.int64)
        rows.append((nb * tiles / 512.0, ms, tf, int(s[3] - s[0])))
    x = np.array([r[0] for r in rows]); y = np.array([r[1] for r in rows])
    b, a = np.polyfit(x, y, 1)
    print(f"M = N = {M}, K = {K}: " + "; ".join(f"{r[0]:.0f} tiles/slot {r[1] * 1e3:.0f} us ({r[2]:.1f} TF, tile stamp {r[3]} cyc)" for r in rows))
    print(f"    fit: {a * 1e3:.1f} us + {b * 1e3:.2f} us per tile and slot; pipe time of one tile pair at 2.1 GHz: {2 * (K // 16) * 4096 / 2.1e3:.1f} us", flush=True)
