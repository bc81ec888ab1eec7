"""Yardstick only (nothing in the product links a BLAS): what AMD's own tuned float64 GEMM kernels (rocBLAS / hipBLASLt
through torch.bmm / torch.baddbmm) reach on this GPU at the shapes of the evaluation's products - an upper mark for
what a 128 x 128-tile MFMA kernel can be expected to sustain next to the 78.6 TFLOP/s datasheet figure.
Run on the GPU box: python tools/probe_vendor_dgemm.py"""
import time

import torch

assert torch.cuda.is_available()
dev = torch.device("cuda", 0)
print(torch.__version__, torch.cuda.get_device_name(0))


def bench(fn, flops, iters=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return flops / dt / 1e12, dt * 1e3


for (nb, m, n, k) in [(256, 1024, 1024, 1024), (64, 2048, 2048, 2048), (512, 512, 512, 512), (128, 2048, 2048, 128),
                      (512, 1024, 1024, 512), (8, 8192, 8192, 8192)]:
    A = torch.randn(nb, m, k, dtype=torch.float64, device=dev)
    B = torch.randn(nb, k, n, dtype=torch.float64, device=dev)
    C = torch.zeros(nb, m, n, dtype=torch.float64, device=dev)
    fl = 2.0 * nb * m * n * k
    Bt = B.transpose(1, 2).contiguous()   # [nb, n, k]: the "NT" operand
    At = A.transpose(1, 2).contiguous()   # [nb, k, m]: the "TN" operand
    for name, fn in [("NN bmm", lambda: torch.bmm(A, B, out=C)),
                     ("NT bmm", lambda: torch.bmm(A, Bt.transpose(1, 2), out=C)),
                     ("TN baddbmm beta=1", lambda: torch.baddbmm(C, At.transpose(1, 2), B, alpha=1e-3, out=C))]:
        try:
            tf, ms = bench(fn, fl)
            print(f"batch {nb:4d}  {m}x{n}x{k:<5d} {name:18s} {tf:6.1f} TFLOP/s  {ms:8.2f} ms", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"batch {nb:4d}  {m}x{n}x{k} {name}: {e!r}", flush=True)
    del A, B, C, At, Bt
    torch.cuda.empty_cache()
