"""How the lower-only M'M product (bench mode 1) and the full GEMM scale with N at constant total work."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _probes_lib  # noqa: F401  (diagnostic build: tools/libwv_probes.so)
from waveome_amd.engine import Context
ctx = Context(0)
for N, nb in [(1024, 1024), (2048, 256), (4096, 64), (8192, 16), (16384, 4)]:
    for mode, name in [(1, "lauum"), (0, "full"), (2, "syrk")]:
        nbb = nb if mode else max(1, nb // 4)
        tf, ms = ctx.gemm_bench(N, N, N, a_kcontig=False, b_kcontig=False, bn=128 + 1000 * mode, nbatch=nbb, iters=2)
        print(f"f64 N={N:6d} nb={nbb:5d} {name:6s}: {tf:6.1f} TFLOP/s {ms:9.3f} ms", flush=True)
