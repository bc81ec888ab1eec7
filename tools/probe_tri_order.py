"""L2 experiment: K_y^-1 = M'M (and the trailing update) with the strictly-lower tiles enumerated row by row (shipped) or
in 4 x 4 super-blocks (WV_TRI_ORDER=1; probes build only), one process per setting."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _probes_lib  # noqa: F401,E402
from waveome_amd.engine import Context  # noqa: E402

c = Context(0)
print("WV_TRI_ORDER =", os.environ.get("WV_TRI_ORDER", "0"))
for (M, nb, mode, name) in [(2048, 256, 1, "M'M 2048"), (2048, 512, 1, "M'M 2048"), (1024, 512, 2, "SYRK 1024 beta=1"),
                            (4096, 64, 1, "M'M 4096")]:
    for rep in range(2):
        tf, ms = c.gemm_bench(M, M, M, False, False, 128 + 1000 * mode, nb, 3)
        print(f"{name:18s} nb{nb}: {tf:6.1f} TF/s {ms:9.3f} ms", flush=True)
