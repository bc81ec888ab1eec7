"""Per-tile fixed cost of the float64 GEMM kernel: full products (every tile the same contraction length K) at
M = N = 2048, 64 matrices, for a range of K - the rate follows peak * K / (K + K0)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _probes_lib  # noqa: F401  (diagnostic build: tools/libwv_probes.so)
from waveome_amd.engine import Context
ctx = Context(0)
for beta_mode, name in [(0, "beta=0"), (5, "beta=1")]:
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        tf, ms = ctx.gemm_bench(2048, 2048, K, a_kcontig=False, b_kcontig=False, bn=128 + 1000 * beta_mode, nbatch=64, iters=3)
        print(f"f64 2048x2048 K={K:5d} {name}: {tf:6.1f} TFLOP/s {ms:8.3f} ms  -> per-tile time {ms * 1e3 / (64 * 256 / 512):7.2f} us (512 tile slots)", flush=True)
