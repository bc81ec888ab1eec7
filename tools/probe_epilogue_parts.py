"""What the hand-scheduled kernels' epilogue costs inside full launches (probes build, results wrong by construction):
WV_GEMM_DBG bits 2 = no epilogue, 512 = no staging of the accumulators, 1 = no C stores."""
import os, sys
sys.path.insert(0, ".")
import _probes_lib  # noqa: F401
from waveome_amd.engine import Context
c = Context(0)
shapes = [(1024, 1024, 256, 0, 0, "full 1024^3 x 256"), (512, 512, 512, 0, 0, "full 512^3 x 512"), (256, 256, 1024, 0, 0, "full 256^3 x 1024"),
          (1024, 1024, 512, 1, 3, "S-like"), (1024, 1024, 512, 0, 6, "M21-like"), (2048, 2048, 512, 0, 1, "M'M"), (1024, 512, 512, 0, 2, "update K=512")]
for dbg, name in ((0, "as shipped"), (1, "no C stores"), (512, "no staging"), (513, "no staging, no stores"), (2, "no epilogue")):
    os.environ["WV_GEMM_DBG"] = str(dbg); os.environ["WV_GEMM_ASM"] = "1"
    row = []
    for (M, K, nb, akc, mode, nm) in shapes:
        tf, ms = c.gemm_bench(M, M, K, bool(akc), False, 128 + 1000 * mode, nb, 3)
        row.append(f"{nm} {tf:5.1f}")
    print(f"{name:24s} | " + " | ".join(row), flush=True)
