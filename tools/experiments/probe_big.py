"""How profiles/r03_probe_big_v*.txt were made: the 256 x 128 one-wave-per-SIMD kernel (wv_gemm3_one_wave_per_simd.hip in
this directory, when it was routed from launch_gemm) against the 128 x 128 kernels on the products of a 512-fit round at
N = 2048, one process per setting:  WV_BIG=0 python probe_big.py ; WV_BIG=1 python probe_big.py   (probes build)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _probes_lib  # noqa: F401,E402
from waveome_amd.engine import Context  # noqa: E402

c = Context(0)
print("WV_BIG =", os.environ.get("WV_BIG", "1 (default)"))
for (M, N, K, nb, mode, name) in [(2048, 2048, 2048, 256, 1, "M'M 2048 (lower, k >= row)"),
                                   (1024, 1024, 1024, 512, 2, "SYRK 1024 beta=1 (lower)"),
                                   (1024, 1024, 1024, 512, 3, "T-like 1024 (k < n0 + BN)"),
                                   (1024, 1024, 1024, 512, 6, "M21-like 1024 (k < m0 + BM)"),
                                   (1024, 1024, 1024, 256, 0, "full 1024"),
                                   (512, 512, 512, 512, 2, "SYRK 512 beta=1 (lower)"),
                                   (512, 512, 512, 512, 0, "full 512")]:
    tf, ms = c.gemm_bench(M, N, K, False, False, 128 + 1000 * mode, nb, 3)
    print(f"{name:30s} nb{nb}: {tf:6.1f} TF/s {ms:9.3f} ms", flush=True)
