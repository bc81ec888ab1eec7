"""Where a GEMM tile's time goes: the float64 kernel with parts switched off (WV_GEMM_DBG bits: 1 no C stores,
2 no epilogue, 4 no operand loads in the K loop, 8 no MFMAs).  Results are wrong by construction - timing only.
Run once per setting: WV_GEMM_DBG=<bits> python tools/probe_gemm_parts.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _probes_lib  # noqa: F401  (diagnostic build: tools/libwv_probes.so)
from waveome_amd.engine import Context
c = Context(0)
print("WV_GEMM_DBG =", os.environ.get("WV_GEMM_DBG", "0"))
for (M, K, nb, akc, bkc, mode, name) in [(2048, 2048, 128, 0, 0, 1, "lauum 2048"), (1024, 1024, 256, 0, 0, 2, "syrk 1024 beta=1"),
                                         (1024, 1024, 256, 0, 0, 0, "full 1024"), (1024, 1024, 256, 0, 0, 3, "trimmed 1024"),
                                         (2048, 128, 64, 0, 0, 0, "full 2048 K=128")]:
    tf, ms = c.gemm_bench(M, M, K, bool(akc), bool(bkc), 128 + 1000 * mode, nb, 3)
    print(f"{name:18s} nb{nb}: {tf:6.1f} TF/s {ms * 1e3:9.1f} us", flush=True)
